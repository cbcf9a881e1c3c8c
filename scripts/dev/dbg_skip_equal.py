#!/usr/bin/env python3
"""Where do the gated (SMD_BWD_SKIP=2) and plain (=0) row loops of the fused backward differ?  (GPU box)
usage: dbg_skip_equal.py [b h w n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from slowtv_monodepth_amd import functional as F
b, h, w, n = (int(v) for v in sys.argv[1:5]) if len(sys.argv) >= 5 else (4, 96, 320, 2)
gen = torch.Generator(device='cuda').manual_seed(3)
imgs = torch.rand(b, 3, h, w, device='cuda', generator=gen)
K = torch.tensor([[0.58*w, 0, 0.5*w, 0], [0, 1.92*h, 0.5*h, 0], [0, 0, 1, 0], [0, 0, 0, 1]], device='cuda')[None].repeat(b, 1, 1)
T0 = torch.eye(4, device='cuda').repeat(n, b, 1, 1); T0[..., :3, 3] = 0.05*torch.randn(n, b, 3, device='cuda', generator=gen)
d0 = [0.05 + 0.9*torch.rand(b, 1, h >> s, w >> s, device='cuda', generator=gen) for s in range(4)]
flags = F.recon_flags('ssim', True, True)
supp = torch.rand(n, b, 3, h, w, device='cuda', generator=gen)

def step(mode, fused=True):
    os.environ['SMD_BWD_SKIP'] = mode
    d = [v.clone().requires_grad_(True) for v in d0]; T = T0.clone().requires_grad_(True)
    if fused:
        loss, _, sel, _, dep = F.image_recon_fused_disp(d, imgs, supp, T, K, flags=flags, min_depth=0.1, max_depth=100, seed=2, want_err=False)
        loss.backward()
        return sel, [v.grad for v in d] + [T.grad]
    depth_up, _ = F.disp_to_depth(d, (h, w), 0.1, 100)
    depth_up.retain_grad()
    loss, _, sel, _ = F.image_recon_fused(depth_up, imgs, supp, T, K, flags=flags, seed=2, want_err=False)
    loss.backward()
    return sel, [depth_up.grad, T.grad]

for fused in (False, True):
    s2, g2 = step('2', fused); s0, g0 = step('0', fused)
    print(f'fused K0: {fused}; sel equal {torch.equal(s2, s0)}')
    for k, (x, y) in enumerate(zip(g2, g0)):
        ne = (x != y) | (torch.isnan(x) != torch.isnan(y))
        print(f'  grad {k} shape {tuple(x.shape)}: differing {int(ne.sum())}, nan {int(torch.isnan(x).sum())}/{int(torch.isnan(y).sum())}, max abs diff {(x - y).abs().nan_to_num().max().item():.3e} of {y.abs().nan_to_num().max().item():.3e}')
        if ne.any() and x.ndim >= 4:
            idx = ne.nonzero()
            rows = idx[:, -2].unique().tolist(); cols = idx[:, -1].unique()
            print(f'    rows {rows[:40]}{"..." if len(rows) > 40 else ""}; cols {cols.min().item()}..{cols.max().item()} ({len(cols)} distinct); first {idx[:5].tolist()}')
            i = tuple(idx[0].tolist()); print(f'    values at first: gated {x[i].item():.6e} plain {y[i].item():.6e}')
