#!/usr/bin/env python3
"""Where do SMD_FWD_SHARE=1 and =0 differ? (GPU box)  usage: dbg_share.py [b h w n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from slowtv_monodepth_amd import functional as F
b, h, w, n = (int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (12, 192, 640, 2)
gen = torch.Generator(device='cuda').manual_seed(h*w + n)
imgs = torch.rand(b, 3, h, w, device='cuda', generator=gen); supp = torch.rand(n, b, 3, h, w, device='cuda', generator=gen)
K = torch.tensor([[0.58*w, 0, 0.5*w, 0], [0, 1.92*h, 0.5*h, 0], [0, 0, 1, 0], [0, 0, 0, 1]], device='cuda')[None].repeat(b, 1, 1)
T = torch.eye(4, device='cuda').repeat(n, b, 1, 1); T[..., :3, 3] = 0.05*torch.randn(n, b, 3, device='cuda', generator=gen)
d = [0.05 + 0.9*torch.rand(b, 1, h >> s, w >> s, device='cuda', generator=gen) for s in range(4)]
flags = F.recon_flags('ssim', True, True)
def run(share):
    os.environ['SMD_FWD_SHARE'] = str(share)
    loss, err, sel, _, dep = F.image_recon_fused_disp(d, imgs, supp, T, K, flags=flags, min_depth=0.1, max_depth=100, seed=7, want_err=True)
    torch.cuda.synchronize()
    return loss, err, sel, dep
for rep in range(3):
    l1, e1, s1, d1 = run(1); l0, e0, s0, d0 = run(0)
    print(f'rep {rep}: loss {l1.item():.8f} {l0.item():.8f}; shapes', tuple(e1.shape), tuple(s1.shape), tuple(d1.shape))
    for name, x, y in (('depth', d1, d0), ('err', e1, e0), ('sel', s1, s0)):
        x = x.reshape(4, b, h, w); y = y.reshape(4, b, h, w)
        bad = (x != y)
        print(f'  {name}: {int(bad.sum())} mismatches', end='')
        if bad.any():
            idx = bad.nonzero()
            print('; scales', sorted(set(idx[:, 0].tolist())), 'samples', sorted(set(idx[:, 1].tolist()))[:12], 'rows', sorted(set(idx[:, 2].tolist()))[:24], 'cols', sorted(set(idx[:, 3].tolist()))[:16],
                  'max |diff|', float((x.float() - y.float()).abs().max()))
        else: print()
    l1b, e1b, s1b, d1b = run(1)
    print('  share=1 run-to-run identical:', torch.equal(e1, e1b), torch.equal(s1, s1b), torch.equal(d1, d1b))
