"""One case of tests/test_gpu_fuzz.py under the magnifying glass (GPU box): HIP and the fp32 oracle against the fp64 oracle under the SAME routing.
usage: python tests/fuzz_case.py <seed>"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from test_gpu_fuzz import draw
from conftest import rel_to_max
from oracle import view_synth_oracle as O
from slowtv_monodepth_amd import functional as F

seed = int(sys.argv[1])
b, h, w, n, lows, opts, lo, hi = draw(seed); S = len(lows)
gen = torch.Generator().manual_seed(seed)
imgs = torch.rand(b, 3, h, w, generator=gen)
mix = 0.6*torch.rand(1, generator=gen).item()
supp = mix*imgs[None] + (1 - mix)*torch.rand(n, b, 3, h, w, generator=gen)
disps = {s: 0.05 + 0.9*torch.rand(b, 1, hs, ws, generator=gen) for s, (hs, ws) in enumerate(lows)}
aa = 0.02*torch.randn(n*b, 3, generator=gen); t = 0.1*torch.randn(n*b, 3, generator=gen)
K = torch.tensor([[0.58*w, 0, 0.5*w, 0], [0, 1.92*h, 0.5*h, 0], [0, 0, 1, 0], [0, 0, 0, 1]])[None].repeat(b, 1, 1)
noise = torch.randn(S*b, 1, h, w, generator=gen) if opts['use_automask'] else None
print(f'seed {seed}: b={b} {h}x{w} n={n} pyramid {lows} {opts} depth [{lo}, {hi}]')

def oracle(dtype, force_sel=None):
    dc = {s: d.to(dtype).clone().requires_grad_(True) for s, d in disps.items()}
    Tc = O.T_from_AAt(aa.to(dtype), t.to(dtype)).unflatten(0, (n, b)).clone().requires_grad_(True)
    loss, out = O.loss_path(dc, imgs.to(dtype), supp.to(dtype), Tc, K.to(dtype), min_depth=lo, max_depth=hi, loss_name=opts['loss_name'], use_min=opts['use_min'],
                            use_automask=opts['use_automask'], use_edges=opts['use_edges'], w_smooth=0.1, noise=None if noise is None else noise.to(dtype), force_sel=force_sel)
    loss.backward()
    return loss.detach(), out, [dc[s].grad for s in dc] + [Tc.grad[..., :3, :]]

l32, o32, g32 = oracle(torch.float32)
dg = [d.cuda().requires_grad_(True) for d in disps.values()]
Tg = O.T_from_AAt(aa, t).unflatten(0, (n, b)).cuda().requires_grad_(True)
l_rec, err, sel, _, dep = F.image_recon_fused_disp(dg, imgs.cuda(), supp.cuda(), Tg, K.cuda(), flags=F.recon_flags(opts['loss_name'], opts['use_min'], opts['use_automask']),
                                                   min_depth=lo, max_depth=hi, noise=None if noise is None else noise.cuda(), want_err=True)
l_sm, _, _ = F.disp_smooth_fused(dict(enumerate(dg)), imgs.cuda(), use_edges=opts['use_edges'], want_aux=False)
(l_rec + 0.1*l_sm).backward()
gh = [d.grad.cpu() for d in dg] + [Tg.grad.cpu()[..., :3, :]]
print('flips', (sel.cpu() != o32['full']['sel']).sum().item(), 'loss hip', (l_rec + 0.1*l_sm).item(), 'oracle32', l32.item())
l64, o64, g64 = oracle(torch.float64, force_sel=o32['full']['sel'].flatten(0, 1) if hasattr(o32['full']['sel'], 'flatten') else None)
for k, (a, c, d) in enumerate(zip(gh, g32, g64)):
    name = f'disp_{k}' if k < S else 'T'
    print(f'  {name}: hip vs oracle32 {rel_to_max(a, c):.2e} | hip vs fp64 {rel_to_max(a.double(), d):.2e} | oracle32 vs fp64 {rel_to_max(c.double(), d):.2e} | max |g| {d.abs().max().item():.2e}')
    if k < S:
        e = (a.double() - d).abs(); i = e.argmax().item(); idx = [int(v) for v in torch.unravel_index(torch.tensor(i), e.shape)]
        print(f'     worst at {idx}: hip {a.flatten()[i].item():.6e} fp32 {c.flatten()[i].item():.6e} fp64 {d.flatten()[i].item():.6e}')

# ---- where does disp_0's gradient differ, and what do the sample coordinates look like there? (scale 0, fp64 geometry)
e = (gh[0].double() - g64[0]).abs()
thr = 1e-4*g64[0].abs().max()
bad = (e > thr).nonzero()
print(f'pixels of disp_0 off by more than 1e-4 of the max: {len(bad)} of {e.numel()}')
hs, ws = lows[0]
if (hs, ws) == (h, w):
    dep64 = o64['depth_up'][0].detach()
    T64 = O.T_from_AAt(aa.double(), t.double()).unflatten(0, (n, b))
    order = e.flatten().argsort(descending=True)[:8]
    for i in order.tolist():
        bi, _, v, u = [int(x) for x in torch.unravel_index(torch.tensor(i), e.shape)]
        line = f'  ({bi},{v},{u}) |diff| {e.flatten()[i].item():.3e} hip {gh[0].flatten()[i].item():.4e} fp64 {g64[0].flatten()[i].item():.4e} depth {dep64[bi,0,v,u].item():.4f}:'
        for k in range(n):
            sx, sy, z, _ = O.sample_coords(dep64[bi:bi+1], T64[k, bi:bi+1], K[bi:bi+1].double())
            # 3x3 neighbourhood: the SSIM window reaches the neighbours' samples too
            nb = [(sx[0, vv, uu].item(), sy[0, vv, uu].item(), z[0, 0, vv, uu].item()) for vv in range(max(v-1,0), min(v+2,h)) for uu in range(max(u-1,0), min(u+2,w))]
            c = (sx[0, v, u].item(), sy[0, v, u].item(), z[0, 0, v, u].item())
            near = [p for p in nb if abs(p[0]) < 1e-3 or abs(p[0] - (w-1)) < 1e-3 or abs(p[1]) < 1e-3 or abs(p[1] - (h-1)) < 1e-3 or abs(p[2] - 0.1) < 1e-3]
            line += f' k{k} s=({c[0]:.3f},{c[1]:.3f}) z={c[2]:.4f}{" NEAR-EDGE " + str(near) if near else ""};'
        print(line)
