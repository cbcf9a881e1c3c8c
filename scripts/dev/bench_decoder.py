#!/usr/bin/env python3
"""Time the decoder glue kernels at the stage shapes of cfg 2 (b=12, 640x192, ResNet-18 skips) against the ATen chain."""
import torch, torch.nn.functional as TF
import slowtv_monodepth_amd.functional as HF

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/n*1e3

b = 12
stages = [(256, 256, 6, 20), (128, 128, 12, 40), (64, 64, 24, 80), (32, 64, 48, 160), (16, 0, 96, 320)]   # Ca, Cs, h, w of `a`
tot = {'hip_fwd': 0, 'hip_bwd': 0, 'aten_fwd': 0, 'aten_bwd': 0}
for Ca, Cs, h, w in stages:
    a = torch.randn(b, Ca, h, w, device='cuda', requires_grad=True)
    skip = torch.randn(b, Cs, 2*h, 2*w, device='cuda', requires_grad=True) if Cs else None
    c = torch.randn(b, Ca, 2*h, 2*w, device='cuda', requires_grad=True)
    def aten_up():
        up = TF.interpolate(TF.elu(a), scale_factor=2, mode='nearest')
        return TF.pad(torch.cat((up, skip), 1) if Cs else up, (1, 1, 1, 1), mode='reflect')
    def aten_pad(): return TF.pad(TF.elu(c), (1, 1, 1, 1), mode='reflect')
    rows = []
    for name, f_hip, f_aten, leaves in (('up_cat_pad', lambda: HF.elu_up_cat_pad(a, skip), aten_up, [a] + ([skip] if Cs else [])),
                                        ('elu_pad', lambda: HF.elu_pad(c, None, True), aten_pad, [c])):
        o_h, o_a = f_hip(), f_aten()
        g = torch.randn_like(o_h)
        t_hf, t_af = timeit(f_hip), timeit(f_aten)
        t_hb = timeit(lambda: torch.autograd.grad(o_h, leaves, g, retain_graph=True))
        t_ab = timeit(lambda: torch.autograd.grad(o_a, leaves, g, retain_graph=True))
        mb = (o_h.numel() + sum(l.numel() for l in leaves))*4/1e6
        rows.append(f'{name}: hip fwd {t_hf:6.1f} bwd {t_hb:6.1f} | aten fwd {t_af:6.1f} bwd {t_ab:6.1f} us | {mb:6.1f} MB -> fwd {mb/t_hf/1e3*1e3:5.2f} GB/ms')
        tot['hip_fwd'] += t_hf; tot['hip_bwd'] += t_hb; tot['aten_fwd'] += t_af; tot['aten_bwd'] += t_ab
    print(f'Ca={Ca:3d} Cs={Cs:3d} {h:3d}x{w:3d}  ' + '  ||  '.join(rows))
print({k: round(v, 1) for k, v in tot.items()})
