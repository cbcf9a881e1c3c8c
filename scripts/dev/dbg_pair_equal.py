#!/usr/bin/env python3
"""Two-supports-per-wave backward (SMD_BWD_PAIR=1) against the one-support-per-wave kernel: bit-equality at n = 2, closeness at n = 4.  (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from slowtv_monodepth_amd import functional as F, _lib
os.environ['SMD_BWD_SKIP'] = '0'
for (b, h, w, n) in ((4, 96, 320, 2), (12, 192, 640, 2), (2, 50, 130, 2), (3, 96, 200, 4), (1, 7, 66, 2)):
    gen = torch.Generator(device='cuda').manual_seed(3)
    imgs = torch.rand(b, 3, h, w, device='cuda', generator=gen)
    K = torch.tensor([[0.58*w, 0, 0.5*w, 0], [0, 1.92*h, 0.5*h, 0], [0, 0, 1, 0], [0, 0, 0, 1]], device='cuda')[None].repeat(b, 1, 1)
    T0 = torch.eye(4, device='cuda').repeat(n, b, 1, 1); T0[..., :3, 3] = 0.05*torch.randn(n, b, 3, device='cuda', generator=gen)
    S = 4 if h >= 48 else 2
    d0 = [0.05 + 0.9*torch.rand(b, 1, max(h >> s, 1), max(w >> s, 1), device='cuda', generator=gen) for s in range(S)]
    flags = F.recon_flags('ssim', True, True)
    supp = (imgs[None] + 0.3*torch.rand(n, b, 3, h, w, device='cuda', generator=gen)).clamp(0, 1)
    def step(pair):
        os.environ['SMD_BWD_PAIR'] = pair
        d = [v.clone().requires_grad_(True) for v in d0]; T = T0.clone().requires_grad_(True)
        loss, _, sel, _, dep = F.image_recon_fused_disp(d, imgs, supp, T, K, flags=flags, min_depth=0.1, max_depth=100, seed=2, want_err=False)
        loss.backward(); torch.cuda.synchronize()
        return [v.grad for v in d] + [T.grad], _lib.lib.smd_last_kernel_variant(1).decode()
    g1, k1 = step('1'); g0, k0 = step('0')
    worst = max(((x - y).abs().max()/y.abs().max().clamp(min=1e-30)).item() for x, y in zip(g1, g0))
    print(f'b={b} {h}x{w} n={n}: [{k1}] vs [{k0}]: bit-equal {all(torch.equal(x, y) for x, y in zip(g1, g0))}, worst rel-to-max {worst:.2e}, finite {all(torch.isfinite(x).all().item() for x in g1)}', flush=True)
