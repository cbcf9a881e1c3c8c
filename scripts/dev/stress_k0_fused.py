#!/usr/bin/env python3
"""Randomised sweep: K0-fused forward + backward (shared ring on) against the K0 kernel followed by the plain fused path, values and
gradients, over shapes / pyramids / strip heights / support counts.  (GPU box)  usage: stress_k0_fused.py [cases] [seed]"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from slowtv_monodepth_amd import functional as F
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
def rel_to_max(a, b): return float((a - b).abs().max()/b.abs().max().clamp_min(1e-30))
bad = 0
for it in range(cases):
    b, n = rng.randint(1, 5), rng.randint(1, 4)
    h, w = rng.randint(8, 100), rng.randint(16, 220)
    if rng.random() < 0.2: h, w = rng.choice([(192, 640), (96, 320)])
    lows = [(max(h >> s, 1), max(w >> s, 1)) for s in range(4)] if rng.random() < 0.6 else [(rng.randint(1, h), rng.randint(1, w)) for _ in range(4)]
    rh = rng.choice([4, 8, 12, 16]); b2 = rng.randint(0, b); rh2 = rng.choice([4, 8])
    for k, v in (('SMD_FWD_RH', rh), ('SMD_BWD_RH', rh), ('SMD_FWD_TAPER_B', b2), ('SMD_BWD_TAPER_B', b2), ('SMD_FWD_TAPER_RH', rh2), ('SMD_BWD_TAPER_RH', rh2)): os.environ[k] = str(v)
    gen = torch.Generator(device='cuda').manual_seed(1000 + it)
    imgs = torch.rand(b, 3, h, w, device='cuda', generator=gen); supp = torch.rand(n, b, 3, h, w, device='cuda', generator=gen)
    K = torch.tensor([[0.58*w, 0, 0.5*w, 0], [0, 1.92*h, 0.5*h, 0], [0, 0, 1, 0], [0, 0, 0, 1]], device='cuda')[None].repeat(b, 1, 1)
    T0 = torch.eye(4, device='cuda').repeat(n, b, 1, 1); T0[..., :3, 3] = 0.05*torch.randn(n, b, 3, device='cuda', generator=gen)
    d0 = [0.05 + 0.9*torch.rand(b, 1, hs, ws, device='cuda', generator=gen) for hs, ws in lows]
    gup = torch.randn(4, b, 1, h, w, device='cuda', generator=gen)
    noise = torch.randn(4*b, 1, h, w, device='cuda', generator=gen)   # same tie-break in both paths
    flags = F.recon_flags('ssim', rng.random() < 0.7, rng.random() < 0.7)
    def run(fused, with_noise):
        d = [v.clone().requires_grad_(True) for v in d0]; T = T0.clone().requires_grad_(True)
        kw = {'noise': noise} if with_noise else {'seed': 5}
        if fused: loss, err, sel, _, dep = F.image_recon_fused_disp(d, imgs, supp, T, K, flags=flags, min_depth=0.1, max_depth=100, want_err=True, **kw)
        else:
            dep, _ = F.disp_to_depth(d, (h, w), 0.1, 100)
            loss, err, sel, _ = F.image_recon_fused(dep, imgs, supp, T, K, flags=flags, want_err=True, **kw)
        (loss + 1e-3*(dep*gup).sum()).backward()
        torch.cuda.synchronize()
        return loss.detach(), err, sel, dep.detach(), [v.grad for v in d], T.grad
    # (1) hot instantiation with the shared ring against the same without it: forward AND backward bit for bit
    os.environ['SMD_FWD_SHARE'] = '1'; la, ea, sa, da, ga, ta = run(True, False)
    os.environ['SMD_FWD_SHARE'] = '0'; lc, ec, sc_, dc, gc, tc = run(True, False)
    os.environ['SMD_FWD_SHARE'] = '1'
    exact = torch.equal(ea, ec) and torch.equal(sa, sc_) and torch.equal(da, dc) and all(torch.equal(x, y) for x, y in zip(ga + [ta], gc + [tc]))
    # (2) against the two-launch path (K0 kernel, then the plain fused operator): depth differs by an ulp (v_rcp vs division), so a tap
    # can sit on the other side of a texel boundary and single gradient entries move by O(1) of their own size: values tight, gradients
    # by the share of entries that moved
    lb, eb, sb, db, gb, tb = run(False, False)
    flips = (sa != sb).float().mean().item()
    ok = exact and torch.allclose(da, db, rtol=1e-6, atol=1e-7) and flips <= 2e-3 and ((ea - eb).abs() > 1e-4).float().mean().item() <= 2e-3 and abs(la.item() - lb.item()) <= 1e-5*abs(lb.item()) + 1e-7
    worst = max(rel_to_max(x, y) for x, y in zip(ga + [ta], gb + [tb]))
    moved = max(((x - y).abs() > 1e-3*y.abs().max()).float().mean().item() for x, y in zip(ga, gb))
    soft = moved <= 2e-3 and rel_to_max(ta, tb) < 5e-2   # reported, not counted: coarse levels have a few dozen entries, one moved tap is several percent of them
    if not ok or not soft:
        bad += 0 if ok else 1
        print(f'{"MISMATCH" if not ok else "note"} case {it}: b={b} n={n} {h}x{w} lows={lows} rh={rh}/{rh2} taper={b2} flags={flags}: exact {exact} flips {flips:.2e} loss {la.item():.8f} {lb.item():.8f} worst grad {worst:.2e} moved {moved:.2e}', flush=True)
    elif it % 10 == 0: print(f'case {it} ok (b={b} n={n} {h}x{w} rh={rh}/{rh2} taper {b2}; flips {flips:.1e}, worst grad {worst:.1e})', flush=True)
print(f'{cases} cases, {bad} mismatches'); sys.exit(1 if bad else 0)
