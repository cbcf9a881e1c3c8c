#!/usr/bin/env python3
"""Exploration for a later round: are MIOpen's NHWC solvers (no layout transposes) faster for this network than the NCHW picks?
Runs cfg 2 with the stock ATen modules only (this library's producer-side kernels off, they are NCHW-only) in both memory formats
with a full MIOpen search (cudnn.benchmark) and prints the steady-state step times."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ['SMD_NO_MIOPEN_DB'] = '1'
os.environ.setdefault('MIOPEN_USER_DB_PATH', '/tmp/miopen_cl_probe'); os.makedirs(os.environ['MIOPEN_USER_DB_PATH'], exist_ok=True)
os.environ.setdefault('MIOPEN_FIND_MODE', '1')
import torch, bench
from slowtv_monodepth_amd.networks import encoders as E
from slowtv_monodepth_amd.synthetic import make_batch
from slowtv_monodepth_amd.train import StepModule, train_steps
from slowtv_monodepth_amd.trainer import MonoDepthModule
torch.backends.cudnn.benchmark = True
E.BatchNormAct2d.fused_enabled = False
wl = dict(bench.WORKLOADS['cfg2']); dev = torch.device('cuda:0')
for cl in (False, True):
    cfg = bench.make_cfg(wl, cl); cfg['trainer']['overlap_nets'] = False
    m = MonoDepthModule(cfg).to(dev)
    for mod in m.modules():
        if hasattr(mod, '_forward_glued'): mod.upsample_mode = 'nearest'   # decoder: plain path is taken when inputs are channels_last (non-contiguous)
    opt = m.configure_optimizers()['optimizer']
    batch = make_batch(wl['b'], wl['h'], wl['w'], wl['supp'], seed=42, device=dev)
    model = StepModule(m)
    t0 = time.perf_counter(); train_steps(model, opt, lambda it: batch, 2); torch.cuda.synchronize()
    print(f'[probe channels_last={cl}] search done in {time.perf_counter() - t0:.0f} s', flush=True)
    t1 = time.perf_counter(); train_steps(model, opt, lambda it: batch, 20); torch.cuda.synchronize()
    print(f'[probe channels_last={cl}] single-stream ATen-only step: {(time.perf_counter() - t1)/20*1e3:.2f} ms', flush=True)
    del m, opt, model
