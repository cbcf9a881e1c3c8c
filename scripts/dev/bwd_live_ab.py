#!/usr/bin/env python3
"""A/B of the fused backward on the selection maps of a REAL training run, same inputs for every variant (GPU box).
Trains the bench's model for `steps` optimiser steps, captures the arguments of the last `loss_path_fused` call, then times forward + backward of
that call alone under: liveness table on / off (knob bwd_live; AB_KNOB=<name> varies another 0/1 knob instead) x plain / gated row loop (SMD_BWD_SKIP).
usage: bwd_live_ab.py [cfg5|cfg4|cfg2] [steps]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from slowtv_monodepth_amd import functional as F, _lib
from slowtv_monodepth_amd.synthetic import make_batch
from slowtv_monodepth_amd.train import StepModule, train_steps
from slowtv_monodepth_amd.trainer import MonoDepthModule
name = sys.argv[1] if len(sys.argv) > 1 else 'cfg5'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
for kv in filter(None, os.environ.get('MB_KNOBS', '').split(',')):   # e.g. MB_KNOBS=bwd_wps=1 (smd_set_knob), applied to every variant
    k_, v_ = kv.split('='); assert _lib.set_knob(k_, int(v_)), f'knob {k_} is not in this build'
wl = dict(bench.WORKLOADS[name])
torch.manual_seed(42)
module = MonoDepthModule(bench.make_cfg(wl)).cuda()
opt = module.configure_optimizers()['optimizer']
batch = make_batch(wl['b'], wl['h'], wl['w'], wl['supp'], seed=42, device='cuda')
model = StepModule(module)
seen = {}
real = F.loss_path_fused
def spy(*a, **kw):
    seen['a'], seen['kw'] = a, kw
    return real(*a, **kw)
F.loss_path_fused = spy
train_steps(model, opt, lambda it: batch, steps)
torch.cuda.synchronize()
F.loss_path_fused = real
a, kw = seen['a'], dict(seen['kw'])
disps = {k: v.detach().clone().requires_grad_(True) for k, v in a[0].items()}
pose = tuple(v.detach().clone().requires_grad_(True) if (v is not None and v.dtype == torch.float32) else v for v in kw['pose'])
kw.update(pose=pose, prepared=None, intrinsics=None)
Ks, K_inv = a[4].detach(), (a[5].detach() if len(a) > 5 and a[5] is not None else kw.pop('K_inv', None))
n = a[2].shape[0]
def once():
    out = real(disps, a[1], a[2], a[3].detach(), Ks, K_inv, **kw)
    out[0].backward()
    return out
sel = once()[3]
print(f'{name} after {steps} steps: automasked {(sel == 255).float().mean().item():.3f} routed {[round((sel == i).float().mean().item(), 4) for i in range(n)]} '
      f'dead waves (table) {[round(v, 3) for v in F.dead_wave_shares(sel, True, n, table_rh=16).tolist()]} (exact) {[round(v, 3) for v in F.dead_wave_shares(sel, True, n).tolist()]}')
ab_knob = os.environ.get('AB_KNOB', 'bwd_live')
iters, rounds = 5, 6          # the variants are INTERLEAVED (a box's clocks drift over the first seconds: whatever is timed first looks slower)
variants = [(skip, live) for skip in ('0', '2') for live in (1, 0)]
times = {v: ([], []) for v in variants}
for _ in range(3): once()
for r in range(rounds):
    for v in (variants if r % 2 == 0 else variants[::-1]):
        os.environ['SMD_BWD_SKIP'] = v[0]; _lib.set_knob(ab_knob, v[1])
        once(); torch.cuda.synchronize()
        for k in (0, 1): _lib.lib.smd_profile_enable(k, iters)
        for _ in range(iters): once()
        torch.cuda.synchronize()
        for k in (0, 1):
            buf = (C.c_float*iters)(); cnt = C.c_int(0)
            _lib.lib.smd_profile_collect(k, buf, iters, C.byref(cnt)); times[v][k].extend(buf[i]*1e3 for i in range(cnt.value))
            _lib.lib.smd_profile_enable(k, 0)
med = lambda x: sorted(x)[len(x)//2]
for v in variants:
    print(f'  row loop {"gated" if v[0] == "2" else "plain"}, {ab_knob} {v[1]}: forward {med(times[v][0]):.1f} us, backward {med(times[v][1]):.1f} us (min {min(times[v][1]):.1f}, {len(times[v][1])} launches)')
