#!/usr/bin/env python3
"""Liveness table on/off: which gradients differ, where (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from slowtv_monodepth_amd import functional as F, _lib
os.environ['SMD_BWD_SKIP'] = sys.argv[1] if len(sys.argv) > 1 else '0'
b, h, w, n, lows, use_min = 3, 96, 320, 4, [(96, 320), (48, 160), (24, 80), (12, 40)], True
gen = torch.Generator(device='cuda').manual_seed(h + w + n)
imgs = torch.rand(b, 3, h, w, device='cuda', generator=gen)
supp = torch.rand(n, b, 3, h, w, device='cuda', generator=gen)
for i in range(n):
    r0, r1 = i*h//n, (i + 1)*h//n
    supp[i, :, :, r0:r1] = (imgs[:, :, r0:r1] + 0.02*torch.randn(b, 3, r1 - r0, w, device='cuda', generator=gen)).clamp(0, 1)
supp[n - 1, :, :, :, w//3: w//3 + 9] = imgs[:, :, :, w//3: w//3 + 9]
imgs[:, :, : h//5, : w//4] = 1.0; supp[:, :, :, : h//5 + 2, : w//4 + 2] = 1.0
K = torch.tensor([[0.58*w, 0, 0.5*w, 0], [0, 1.92*h, 0.5*h, 0], [0, 0, 1, 0], [0, 0, 0, 1]], device='cuda')[None].repeat(b, 1, 1)
T0 = torch.eye(4, device='cuda').repeat(n, b, 1, 1); T0[..., :3, 3] = 0.002*torch.randn(n, b, 3, device='cuda', generator=gen)
d0 = [0.05 + 0.9*torch.rand(b, 1, hs, ws, device='cuda', generator=gen) for hs, ws in lows]
flags = F.recon_flags('ssim', use_min, True)
def run(live):
    _lib.set_knob('bwd_live', live)
    d = [v.clone().requires_grad_(True) for v in d0]; T = T0.clone().requires_grad_(True)
    loss, _, sel, _, dep = F.image_recon_fused_disp(d, imgs, supp, T, K, flags=flags, min_depth=0.1, max_depth=100, seed=5, want_err=False)
    dep.retain_grad()
    loss.backward(); torch.cuda.synchronize()
    return sel, [v.grad for v in d] + [T.grad]
sel, g1 = run(1); _, g0 = run(0)
print('shares', [(sel == i).float().mean().item() for i in range(n)], (sel == 255).float().mean().item())
for k, (x, y) in enumerate(zip(g1, g0)):
    df = (x - y).abs()
    print(k, tuple(x.shape), 'max diff', df.max().item(), 'n diff', int((df > 0).sum()), 'max |y|', y.abs().max().item())
    if df.max() > 0 and x.ndim == 4 and x.shape[-1] > 8:
        idx = (df > 0).nonzero()
        print('   first diffs at', idx[:6].tolist(), ' rows', sorted(set(idx[:, 2].tolist()))[:20], 'cols', sorted(set(idx[:, 3].tolist()))[:30])
    elif df.max() > 0: print(x.flatten()[:32].tolist(), y.flatten()[:32].tolist())
