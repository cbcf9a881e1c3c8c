#!/usr/bin/env python3
"""How long does the HOST need to enqueue one training step?  Run the cfg-2 step at batch 1 and 64x96 (GPU work negligible):
the step time is then the launch/dispatch cost, i.e. the floor below which a faster GPU path cannot push the step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from slowtv_monodepth_amd.synthetic import make_batch
from slowtv_monodepth_amd.train import StepModule, train_steps
from slowtv_monodepth_amd.trainer import MonoDepthModule
wl = dict(bench.WORKLOADS['cfg2'])
dev = torch.device('cuda:0')
module = MonoDepthModule(bench.make_cfg(wl, False)).to(dev)
opt = module.configure_optimizers()['optimizer']
model = StepModule(module)
for b, h, w in ((1, 64, 96), (12, 192, 640)):
    batch = make_batch(b, h, w, wl['supp'], seed=1, device=dev)
    train_steps(model, opt, lambda it: batch, 5); torch.cuda.synchronize()
    t0 = time.perf_counter(); train_steps(model, opt, lambda it: batch, 20); t_enq = time.perf_counter() - t0
    torch.cuda.synchronize(); t_all = time.perf_counter() - t0
    print(f'b={b} {h}x{w}: host enqueue {t_enq/20*1e3:.2f} ms/step, step {t_all/20*1e3:.2f} ms')
