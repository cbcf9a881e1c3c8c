import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from slowtv_monodepth_amd import functional as F
from slowtv_monodepth_amd.synthetic import make_batch
from slowtv_monodepth_amd.train import StepModule, train_steps
from slowtv_monodepth_amd.trainer import MonoDepthModule
wl = bench.WORKLOADS['cfg2']
module = MonoDepthModule(bench.make_cfg(wl)).cuda(); opt = module.configure_optimizers()['optimizer']
batch = make_batch(wl['b'], wl['h'], wl['w'], wl['supp'], seed=42, device='cuda')
n = [0]; real = F.inv_intrinsics
def spy(K):
    n[0] += 1; print('inv_intrinsics call', n[0], id(K), K._version, flush=True); return real(K)
F.inv_intrinsics = spy
train_steps(StepModule(module), opt, lambda it: batch, 4)
torch.cuda.synchronize(); print('calls in 4 steps:', n[0], 'cache', type(module.backend.__dict__.get('_kinv_cache')))
