#!/usr/bin/env python3
"""Randomised equivalence sweep of the forward's shared target ring (SMD_FWD_SHARE=1) against the per-wave loads (=0): shapes, strip
heights, tapers, pyramids, support counts.  (GPU box)  usage: stress_share.py [cases] [seed]"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from slowtv_monodepth_amd import functional as F
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
flags_all = [F.recon_flags('ssim', True, True), F.recon_flags('ssim', True, False), F.recon_flags('ssim', False, True), F.recon_flags('ssim', False, False)]
bad = 0
for it in range(cases):
    b, n = rng.randint(1, 6), rng.randint(1, 4)
    h, w = rng.randint(4, 90), rng.randint(8, 210)
    if rng.random() < 0.15: h, w = rng.choice([(192, 640), (96, 320), (128, 416)]), None
    if w is None: h, w = h
    if rng.random() < 0.5: lows = [(max(h >> s, 1), max(w >> s, 1)) for s in range(4)]
    else: lows = [(rng.randint(1, h), rng.randint(1, w)) for _ in range(4)]
    rh = rng.choice([4, 8, 12, 16, 20, 32]); b2 = rng.randint(0, b); rh2 = rng.choice([4, 8, 12])
    os.environ['SMD_FWD_RH'] = str(rh); os.environ['SMD_FWD_TAPER_B'] = str(b2); os.environ['SMD_FWD_TAPER_RH'] = str(rh2)
    gen = torch.Generator(device='cuda').manual_seed(it)
    imgs = torch.rand(b, 3, h, w, device='cuda', generator=gen); supp = torch.rand(n, b, 3, h, w, device='cuda', generator=gen)
    K = torch.tensor([[0.58*w, 0, 0.5*w, 0], [0, 1.92*h, 0.5*h, 0], [0, 0, 1, 0], [0, 0, 0, 1]], device='cuda')[None].repeat(b, 1, 1)
    T = torch.eye(4, device='cuda').repeat(n, b, 1, 1); T[..., :3, 3] = 0.05*torch.randn(n, b, 3, device='cuda', generator=gen)
    d = [0.05 + 0.9*torch.rand(b, 1, hs, ws, device='cuda', generator=gen) for hs, ws in lows]
    flags = rng.choice(flags_all)
    out = []
    for share in (1, 0):
        os.environ['SMD_FWD_SHARE'] = str(share)
        loss, err, sel, _, dep = F.image_recon_fused_disp(d, imgs, supp, T, K, flags=flags, min_depth=0.1, max_depth=100, seed=it, want_err=True)
        torch.cuda.synchronize()
        out.append((loss, err, sel, dep))
    (l1, e1, s1, d1), (l0, e0, s0, d0) = out
    ok = torch.equal(e1, e0) and torch.equal(s1, s0) and torch.equal(d1, d0) and abs(l1.item() - l0.item()) <= 1e-6*abs(l0.item())
    if not ok:
        bad += 1
        print(f'MISMATCH case {it}: b={b} n={n} h={h} w={w} lows={lows} rh={rh} b2={b2} rh2={rh2} flags={flags}: err {int((e1 != e0).sum())} sel {int((s1 != s0).sum())} depth {int((d1 != d0).sum())} loss {l1.item()} {l0.item()}', flush=True)
    elif it % 20 == 0: print(f'case {it} ok (b={b} n={n} {h}x{w} rh={rh}/{rh2} taper {b2})', flush=True)
print(f'{cases} cases, {bad} mismatches')
sys.exit(1 if bad else 0)
