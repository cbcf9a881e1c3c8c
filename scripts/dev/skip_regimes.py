#!/usr/bin/env python3
"""Fused backward with and without dead-row skipping against SYNTHETIC selection maps of a given structure (the backward only reads
`sel`; the gradients are meaningless here, the timing is not).  (GPU box)  usage: skip_regimes.py [cfg2|cfg4]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from slowtv_monodepth_amd import functional as F, _lib
from slowtv_monodepth_amd.synthetic import make_batch
name = sys.argv[1] if len(sys.argv) > 1 else 'cfg2'
b, h, w, supp = {'cfg2': (12, 192, 640, (-1, 1)), 'cfg4': (12, 384, 640, (-1, 1))}[name]
S, n, iters = 4, len(supp), 12
_, y, _ = make_batch(b, h, w, supp, seed=42, device='cuda')
g = torch.Generator(device='cuda').manual_seed(0)
import torch.nn.functional as Fn
low = [0.2 + 0.6*torch.rand(b, 1, 4, 10, device='cuda', generator=g) for _ in range(S)]
disps0 = [Fn.interpolate(low[s], size=(h >> s, w >> s), mode='bilinear', align_corners=False) + 0.01*torch.rand(b, 1, h >> s, w >> s, device='cuda', generator=g) for s in range(S)]
T0 = torch.eye(4, device='cuda').repeat(n, b, 1, 1); T0[..., :3, 3] = 0.05*torch.randn(n, b, 3, device='cuda', generator=g)
flags = F.recon_flags('ssim', True, True)

def make_sel(kind, masked):
    """kind: 'pixel' (per-pixel random), 'tile' (per (row, 60-column tile) random), 'blob' (32 x 120 blocks); `masked` = share of units the automask takes"""
    if kind == 'pixel': gh, gw, rh_, rw_ = h, w, 1, 1
    elif kind == 'tile': gh, gw, rh_, rw_ = h, (w + 59)//60, 1, 60
    else: gh, gw, rh_, rw_ = (h + 31)//32, (w + 119)//120, 32, 120
    u = torch.rand(S, b, gh, gw, device='cuda', generator=g)
    v = torch.where(u < masked, torch.full_like(u, 255), torch.where(u < masked + (1 - masked)/2, torch.zeros_like(u), torch.ones_like(u))).to(torch.uint8)
    return v.repeat_interleave(rh_, 2).repeat_interleave(rw_, 3)[:, :, :h, :w].reshape(S, b, 1, h, w).contiguous()

def time_bwd(sel_syn, skip):
    os.environ['SMD_BWD_SKIP'] = str(skip)
    for k in range(5): _lib.lib.smd_profile_enable(k, iters)
    for it in range(iters + 2):
        d = [v.clone().requires_grad_(True) for v in disps0]; T = T0.clone().requires_grad_(True)
        loss, _, sel, _, _ = F.image_recon_fused_disp(d, y['imgs'], y['supp_imgs'], T, y['K'], flags=flags, min_depth=0.1, max_depth=100, seed=1, want_err=False)
        sel.data.copy_(sel_syn)
        loss.backward()
    torch.cuda.synchronize()
    buf = (C.c_float*iters)(); k = C.c_int(0)
    _lib.lib.smd_profile_collect(1, buf, iters, C.byref(k)); v = sorted(buf[i] for i in range(k.value))
    return v[len(v)//2]*1e3

print(f'{name}: fused backward, HIP events around the kernel, median of {iters}; "skippable" = mean of functional.dead_tile_shares of the synthetic map')
for kind in ('pixel', 'tile', 'blob'):
    for masked in (0.0, 0.2, 0.4, 0.6, 0.8, 0.95, 1.0):
        sel_syn = make_sel(kind, masked)
        dead = float(F.dead_tile_shares(sel_syn, True, n).mean())
        t2, t0 = time_bwd(sel_syn, 2), time_bwd(sel_syn, 0)
        print(f'  {kind:5s} masked {masked:4.2f}  skippable {dead:5.3f}   skipping {t2:6.1f} us   plain {t0:6.1f} us   ratio {t2/t0:5.3f}', flush=True)
