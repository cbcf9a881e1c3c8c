#!/usr/bin/env python3
"""bwd_wps = 1 vs min(n, 4): where do the gradients differ (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from slowtv_monodepth_amd import functional as F, _lib
CASES = [(3, 96, 320, 4, [(96, 320), (48, 160), (24, 80), (12, 40)]), (2, 50, 130, 2, [(50, 130), (25, 65)]), (2, 64, 200, 3, [(64, 200), (32, 100), (16, 50)]), (12, 192, 640, 2, [(192, 640), (96, 320), (48, 160), (24, 80)])]
for skip in ('0', '2'):
  for (b, h, w, n, lows) in CASES:
    os.environ['SMD_BWD_SKIP'] = skip
    gen = torch.Generator(device='cuda').manual_seed(h + w + n)
    imgs = torch.rand(b, 3, h, w, device='cuda', generator=gen)
    supp = torch.rand(n, b, 3, h, w, device='cuda', generator=gen)
    for i in range(n):
        r0, r1 = i*h//n, (i + 1)*h//n
        supp[i, :, :, r0:r1] = (imgs[:, :, r0:r1] + 0.02*torch.randn(b, 3, r1 - r0, w, device='cuda', generator=gen)).clamp(0, 1)
    K = torch.tensor([[0.58*w, 0, 0.5*w, 0], [0, 1.92*h, 0.5*h, 0], [0, 0, 1, 0], [0, 0, 0, 1]], device='cuda')[None].repeat(b, 1, 1)
    T0 = torch.eye(4, device='cuda').repeat(n, b, 1, 1); T0[..., :3, 3] = 0.002*torch.randn(n, b, 3, device='cuda', generator=gen)
    d0 = [0.05 + 0.9*torch.rand(b, 1, hs, ws, device='cuda', generator=gen) for hs, ws in lows]
    flags = F.recon_flags('ssim', True, False)
    def run(wps, live):
        _lib.set_knob('bwd_wps', wps); _lib.set_knob('bwd_live', live)
        d = [v.clone().requires_grad_(True) for v in d0]; T = T0.clone().requires_grad_(True)
        loss, _, sel, _, dep = F.image_recon_fused_disp(d, imgs, supp, T, K, flags=flags, min_depth=0.1, max_depth=100, seed=5, want_err=False)
        loss.backward(); torch.cuda.synchronize()
        return [v.grad for v in d] + [T.grad], sel
    for live in (0, 1):
        g1, sel = run(1, live); gn, _ = run(min(n, 4), live)
        out = []
        for k, (x, y) in enumerate(zip(g1, gn)):
            df = (x - y).abs(); out.append(f'#{k}: {df.max().item()/y.abs().max().item():.1e} ({int((df > 0).sum())})')
        print(f'skip={skip} b={b} {h}x{w} n={n} live={live}: rel-to-max diff (elements) ' + '  '.join(out))
