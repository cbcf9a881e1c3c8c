#!/usr/bin/env python3
"""cfg 4's loss path (384x640, b = 12, two supports, learned intrinsics) on LIVE selection maps (GPU box).
The bench's cfg 4 collapses: with randomly initialised learned intrinsics > 99 % of the pixels are auto-masked from the second optimiser step on, so what
it times is the all-masked floor of the backward.  Here the same operator runs on frames / disparities / poses whose masks are alive — the generator of the
BASELINE-resolution reference fixtures (tests/golden/exact_inputs.py + the motion of make_golden.py: a camera translation that roughly explains the frames'
shifts), 12 samples — with both row loops pinned, interleaved.  usage: cfg4_live_masks.py [n_supports] [batch] [height]"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import torch
from exact_inputs import frame_shifts, make_inputs_exact
from slowtv_monodepth_amd import functional as F, _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
for kv in filter(None, os.environ.get('MB_KNOBS', '').split(',')):   # e.g. MB_KNOBS=bwd_wps=1 (smd_set_knob)
    k_, v_ = kv.split('='); assert _lib.set_knob(k_, int(v_)), f'knob {k_} is not in this build'
b, h, w, scales = (int(sys.argv[2]) if len(sys.argv) > 2 else 12), (int(sys.argv[3]) if len(sys.argv) > 3 else 384), 640, [0, 1, 2, 3]
inp = make_inputs_exact(2025, b, h, w, n, scales)
g = torch.Generator().manual_seed(2026)
sh = torch.tensor(frame_shifts(n), dtype=torch.float32)
t0 = torch.stack([-sh[:, 0]*1.2/(0.58*w), -sh[:, 1]*1.2/(1.92*h), torch.zeros(n)], -1)
aa = (0.001*torch.randn(n, b, 3, generator=g)).flatten(0, 1).cuda().requires_grad_(True)
t = (t0[:, None] + 0.001*torch.randn(n, b, 3, generator=g)).flatten(0, 1).cuda().requires_grad_(True)
fs = (torch.tensor([0.58, 1.92])[None].repeat(b, 1)*(1 + 0.1*torch.randn(b, 2, generator=g))).cuda().requires_grad_(True)
cs = (0.5 + 0.05*torch.randn(b, 2, generator=g)).cuda().requires_grad_(True)
imgs, supp = inp['imgs'].cuda(), inp['supp_imgs'].cuda()
disps = {s: inp['disp'][s].cuda().requires_grad_(True) for s in scales}
flags = F.recon_flags('ssim', True, True)
def once():
    Ts = F.pose_matrices(aa, t).unflatten(0, (n, b))
    K, K_inv = F.intrinsics(fs, cs, (h, w))
    out = F.loss_path_fused(disps, imgs, supp, Ts, K, K_inv, pose=(aa, t, None), intrinsics=(fs, cs), flags=flags, min_depth=0.1, max_depth=100, seed=1)
    out[0].backward()
    return out
sel = once()[3]
print(f'{h}x640 b={b} n={n}, learned K, live masks: automasked {(sel == 255).float().mean().item():.3f} routed {[round((sel == i).float().mean().item(), 3) for i in range(n)]} '
      f'dead waves (table) {[round(v, 3) for v in F.dead_wave_shares(sel, True, n, table_rh=16).tolist()]}')
iters, rounds = 5, 6
times = {v: ([], []) for v in ('0', '2')}
for _ in range(3): once()
for r in range(rounds):
    for v in (('0', '2') if r % 2 == 0 else ('2', '0')):
        os.environ['SMD_BWD_SKIP'] = v
        once(); torch.cuda.synchronize()
        for k in (0, 1): _lib.lib.smd_profile_enable(k, iters)
        for _ in range(iters): once()
        torch.cuda.synchronize()
        for k in (0, 1):
            buf = (C.c_float*iters)(); cnt = C.c_int(0)
            _lib.lib.smd_profile_collect(k, buf, iters, C.byref(cnt)); times[v][k].extend(buf[i]*1e3 for i in range(cnt.value))
            _lib.lib.smd_profile_enable(k, 0)
med = lambda x: sorted(x)[len(x)//2]
B = b*h*w*(4*9 + 12*(1 + n))
for v in ('0', '2'):
    f, bw = med(times[v][0]), med(times[v][1])
    print(f'  row loop {"gated" if v == "2" else "plain"}: forward {f:.1f} us = {B/f/1e3/8000:.3f} of 8 TB/s, backward {bw:.1f} us = {B/bw/1e3/8000:.3f} (HIP events; algorithmic bytes {B/1e6:.1f} MB)')
