"""One case of test_random_shapes_with_more_than_four_supports against the fp64 oracle (GPU box).  usage: python tests/fuzz_case_many.py <seed> ..."""
import sys, os, random
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from conftest import rel_to_max
from oracle import view_synth_oracle as O
from slowtv_monodepth_amd import functional as F

for seed in [int(v) for v in sys.argv[1:]]:
    r = random.Random(700 + seed)
    b, h, w, n, S = r.choice([1, 2]), r.randint(3, 60), r.randint(3, 140), r.randint(5, 8), r.choice([1, 2, 3])
    use_min, use_auto = r.random() < 0.7, r.random() < 0.6
    gen = torch.Generator().manual_seed(100 + seed)
    imgs = torch.rand(b, 3, h, w, generator=gen)
    mix = 0.5*torch.rand(1, generator=gen).item()
    supp = mix*imgs[None] + (1 - mix)*torch.rand(n, b, 3, h, w, generator=gen)
    depth = 1 + 10*torch.rand(S, b, 1, h, w, generator=gen)
    aa = 0.02*torch.randn(n*b, 3, generator=gen); t = 0.2*torch.randn(n*b, 3, generator=gen)
    K = torch.tensor([[0.58*w, 0, 0.5*w, 0], [0, 1.92*h, 0.5*h, 0], [0, 0, 1, 0], [0, 0, 0, 1]])[None].repeat(b, 1, 1)
    noise = torch.randn(S*b, 1, h, w, generator=gen)
    print(f'seed {seed}: b={b} {h}x{w} n={n} S={S} min={use_min} automask={use_auto}')
    def orc(dt, force=None):
        d_c = depth.to(dt).clone().requires_grad_(True); T_c = O.T_from_AAt(aa.to(dt), t.to(dt)).unflatten(0, (n, b)).clone().requires_grad_(True)
        loss, _, full = O.image_recon({s: d_c[s] for s in range(S)}, imgs.to(dt), supp.to(dt), T_c, K.to(dt), 'ssim', use_min, use_auto, noise=noise.to(dt), force_sel=force)
        loss.backward(); return loss.detach(), full, d_c.grad, T_c.grad[..., :3, :]
    l32, f32, gd32, gT32 = orc(torch.float32)
    d_g = depth.cuda().requires_grad_(True); T_g = O.T_from_AAt(aa, t).unflatten(0, (n, b)).cuda().requires_grad_(True)
    loss, err, sel, _ = F.image_recon_fused(d_g, imgs.cuda(), supp.cuda(), T_g, K.cuda(), flags=F.recon_flags('ssim', use_min, use_auto), noise=noise.cuda())
    loss.backward()
    l64, f64, gd64, gT64 = orc(torch.float64, force=f32['sel'].flatten(0, 1))
    gdh, gTh = d_g.grad.cpu(), T_g.grad.cpu()[..., :3, :]
    print(f'  flips {(sel.cpu() != f32["sel"]).sum().item()}; loss hip {loss.item():.8f} o32 {l32.item():.8f} o64 {l64.item():.8f}')
    print(f'  depth: hip-o32 {rel_to_max(gdh, gd32):.2e} hip-o64 {rel_to_max(gdh.double(), gd64):.2e} o32-o64 {rel_to_max(gd32.double(), gd64):.2e}')
    print(f'  T    : hip-o32 {rel_to_max(gTh, gT32):.2e} hip-o64 {rel_to_max(gTh.double(), gT64):.2e} o32-o64 {rel_to_max(gT32.double(), gT64):.2e}')
    eT = (gTh.double() - gT64).abs()/gT64.abs().max()
    print('  T error per support (max over samples/entries):', [f'{eT[k].max().item():.1e}' for k in range(n)])
    # the worst element of d loss / d depth and where every support samples there (fp64 geometry): a coordinate within ~1e-5 of an integer?
    e = (gdh.double() - gd64).abs(); i = int(e.argmax()); s_, bi, _, v, u = [int(x) for x in torch.unravel_index(torch.tensor(i), e.shape)]
    T64 = O.T_from_AAt(aa.double(), t.double()).unflatten(0, (n, b))
    line = f'  worst depth element (scale {s_}, sample {bi}, row {v}, col {u}): hip {gdh.flatten()[i].item():.4e} fp64 {gd64.flatten()[i].item():.4e};'
    for k in range(n):
        sx, sy, z, _ = O.sample_coords(depth[s_].double()[bi:bi+1], T64[k, bi:bi+1], K[bi:bi+1].double())
        # the 3x3 window of (v, u): SSIM couples the neighbours' samples to this pixel
        near = []
        for vv in range(max(v-1, 0), min(v+2, h)):
            for uu in range(max(u-1, 0), min(u+2, w)):
                for c in (sx[0, vv, uu].item(), sy[0, vv, uu].item()):
                    if abs(c - round(c)) < 3e-5: near.append(f'{c:.6f}')
        if near: line += f' k{k}: near-integer coordinates in the window {near};'
    print(line)
