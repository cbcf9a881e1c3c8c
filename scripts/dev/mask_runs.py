#!/usr/bin/env python3
"""Row-liveness structure of the selection maps the fused backward sees in the training bench (GPU box).

For every (scale, sample, 64-lane wave window of the backward = 60 columns + 2 halo lanes per side, support): which centre rows have
at least one pixel that routes gradient to that support (L), which rows must be re-synthesised for them (N = L | L<<1 | L>>1), and
how N splits into runs inside the 16-row strips — what a liveness-gated row loop can save, and how often its pipeline restarts.
usage: mask_runs.py [cfg2|cfg4|cfg5] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from slowtv_monodepth_amd import functional as F
from slowtv_monodepth_amd.synthetic import make_batch
from slowtv_monodepth_amd.train import StepModule, train_steps
from slowtv_monodepth_amd.trainer import MonoDepthModule

name = sys.argv[1] if len(sys.argv) > 1 else 'cfg2'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 24
wl = bench.WORKLOADS[name]
torch.manual_seed(42)
module = MonoDepthModule(bench.make_cfg(wl)).cuda().train()
opt = module.configure_optimizers()['optimizer']
model = StepModule(module)
batch = make_batch(wl['b'], wl['h'], wl['w'], wl['supp'], seed=42, device='cuda')
n = len(wl['supp'])
seen = {}
real = F.image_recon_fused_disp
def spy(*a, **kw):
    out = real(*a, **kw); seen['sel'] = out[2]; return out
F.image_recon_fused_disp = spy

def analyse(sel, rh=16):
    S, b, _, h, w = sel.shape
    s4 = sel.reshape(S*b, h, w)
    nsx = (w + 59)//60
    out = []
    for k in range(n):
        bit = (s4 == k)
        # window of wave sx: columns 60*sx-2 .. 60*sx+61 (clipped to the image)
        L = torch.stack([bit[:, :, max(60*sx - 2, 0):min(60*sx + 62, w)].any(-1) for sx in range(nsx)], 1)   # (S*b, nsx, h)
        Lp = torch.nn.functional.pad(L, (1, 1))
        N = Lp[..., :-2] | Lp[..., 1:-1] | Lp[..., 2:]
        # runs of N inside strips of rh rows (+2 halo rows each side are ignored here: interior rows only)
        Ns = N.reshape(S*b, nsx, h//rh, rh)
        starts = (Ns[..., 1:] & ~Ns[..., :-1]).sum(-1) + Ns[..., 0]
        strips_dead = (~Ns.any(-1)).float().mean().item()
        strips_full = Ns.all(-1).float().mean().item()
        out.append(dict(px=bit.float().mean().item(), L=L.float().mean().item(), N=N.float().mean().item(), runs_per_strip=starts.float().mean().item(),
                        strips_dead=strips_dead, strips_full=strips_full))
    masked = (s4 == 255).float().mean().item()
    return masked, out

print(f'{name}: b={wl["b"]} {wl["h"]}x{wl["w"]} n={n}; per support: px = share of pixels selecting it, L = share of live (row, window) units, N = share of rows to re-synthesise,'
      f' runs = runs of N per 16-row strip, dead / full = share of strips with no / only needed rows')
for it in range(steps):
    train_steps(model, opt, lambda i: batch, 1)
    if it in (0, 1, 2, 4, 8, 12, 16, 23, steps - 1):
        torch.cuda.synchronize()
        masked, st = analyse(seen['sel'])
        print(f'  step {it:3d}: automasked {masked:.3f} | ' + ' | '.join(f'k={k}: px {d["px"]:.3f} L {d["L"]:.3f} N {d["N"]:.3f} runs {d["runs_per_strip"]:.2f} dead {d["strips_dead"]:.3f} full {d["strips_full"]:.3f}'
                                                                    for k, d in enumerate(st)), flush=True)
