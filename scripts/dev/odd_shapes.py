import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from slowtv_monodepth_amd.synthetic import make_batch
from slowtv_monodepth_amd.train import StepModule, train_steps
from slowtv_monodepth_amd.trainer import MonoDepthModule
for enc, b, h, w in (('resnet18', 5, 128, 224), ('resnet50', 3, 96, 160), ('convnext_tiny', 3, 160, 96), ('resnet18', 1, 64, 64)):
    wl = dict(bench.WORKLOADS['cfg2']); wl['depth'] = enc
    m = MonoDepthModule(bench.make_cfg(wl, False)).cuda()
    opt = m.configure_optimizers()['optimizer']
    batch = make_batch(b, h, w, wl['supp'], seed=3, device='cuda')
    ls = train_steps(StepModule(m), opt, lambda it: batch, 6); torch.cuda.synchronize()
    print(enc, b, h, w, [round(l.item(), 5) for l in ls])
