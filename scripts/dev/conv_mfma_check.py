#!/usr/bin/env python3
"""The split-bf16 MFMA convolutions (smd_conv3x3_mfma_*) per decoder layer: error against fp64 `conv2d` beside MIOpen's fp32 error, and time of the
forward / data gradient / weight gradient beside MIOpen's (HIP events over the raw C calls, interleaved).  (GPU box.)
usage: conv_mfma_check.py [--quick] [--pieces 3] [--b 12] [--hw 192x640]"""
import argparse, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as TF
from slowtv_monodepth_amd import miopen_tuning  # noqa: F401
from slowtv_monodepth_amd import functional as HF, _lib
from slowtv_monodepth_amd._lib import call

ap = argparse.ArgumentParser()
ap.add_argument('--quick', action='store_true'); ap.add_argument('--pieces', type=int, nargs='*', default=[3, 2]); ap.add_argument('--b', type=int, default=12)
ap.add_argument('--hw', default='192x640'); ap.add_argument('--two-tiles', type=int, default=None); ap.add_argument('--no-time', action='store_true'); ap.add_argument('--layers', nargs='*')
args = ap.parse_args()
H, W = map(int, args.hw.split('x'))
if args.two_tiles is not None: _lib.set_knob('conv_two_tiles', args.two_tiles)
b = args.b
layers = [('up1_0', 16, 16, H, W), ('up0_0', 32, 16, H//2, W//2), ('up1_1', 96, 32, H//2, W//2), ('up0_1', 64, 32, H//4, W//4), ('up1_2', 128, 64, H//4, W//4), ('up0_2', 128, 64, H//8, W//8),
          ('up1_3', 256, 128, H//8, W//8), ('up0_3', 256, 128, H//16, W//16), ('up1_4', 512, 256, H//16, W//16), ('up0_4', 512, 256, H//32, W//32)]
small = [('odd_t', 16, 16, 7, 70), ('odd_u', 32, 16, 9, 33), ('odd_a', 16, 32, 5, 7), ('odd_b', 48, 64, 9, 70), ('odd_c', 32, 32, 33, 65), ('odd_d', 96, 32, 13, 100), ('odd_e', 32, 96, 7, 33), ('odd_f', 64, 128, 4, 20)]
if args.layers: layers = [l for l in layers if l[0] in args.layers]


def rel(a, r): return ((a.double() - r).abs().max()/r.abs().max()).item()


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/n*1e3


def stream(): return torch.cuda.current_stream().cuda_stream


print(f'# b = {b}, image {H}x{W}; error = max |x - fp64| / max |fp64|; times in us (HIP events, 20 calls), TF/s = fp32-equivalent 2 x 9 x C x CO x pixels / time')
for name, C, CO, h, w in (small if args.quick else small + layers):
    B = 2 if name.startswith('odd') else b
    try:
        gen = torch.Generator(device='cuda').manual_seed(C*1000 + CO + h)
        xp = torch.randn(B, C, h + 2, w + 2, device='cuda', generator=gen)
        wt = torch.randn(CO, C, 3, 3, device='cuda', generator=gen)/(3*C**0.5)
        gy = torch.randn(B, CO, h, w, device='cuda', generator=gen)
        R = [t.double().clone().requires_grad_(True) for t in (xp, wt)]
        TF.conv2d(R[0], R[1]).backward(gy.double())
        yr = TF.conv2d(R[0], R[1]).detach()
        M = [t.clone().requires_grad_(True) for t in (xp, wt)]
        ym = TF.conv2d(M[0], M[1]); ym.backward(gy)
        line = f'{name:6s} {C:4d}->{CO:<4d} {h:3d}x{w:<4d} err MIOpen y {rel(ym, yr):.1e} gx {rel(M[0].grad, R[0].grad):.1e} gw {rel(M[1].grad, R[1].grad):.1e}'
        for P in args.pieces:
            L = [t.clone().requires_grad_(True) for t in (xp, wt)]
            y = HF.conv3x3_mfma(L[0], L[1], P); y.backward(gy)
            line += f' | x{P}: y {rel(y, yr):.1e} gx {rel(L[0].grad, R[0].grad):.1e} gw {rel(L[1].grad, R[1].grad):.1e}'
        print(line, flush=True)
        if args.no_time or name.startswith('odd'): continue
        flop = 2.0*B*h*w*C*CO*9
        t_f = timeit(lambda: TF.conv2d(xp, wt))
        t_d = timeit(lambda: torch.ops.aten.convolution_backward(gy, xp, wt, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [True, False, False]))
        t_w = timeit(lambda: torch.ops.aten.convolution_backward(gy, xp, wt, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False]))
        if CO == 16:
            yt = torch.empty(B, CO, h, w, device='cuda'); gxt = torch.empty_like(xp); gwt = torch.empty_like(wt)
            nb = _lib.lib.smd_conv3x3_thin_workspace_bytes(B, C, h, w); wst = torch.empty(nb, device='cuda', dtype=torch.uint8)
            a_f = timeit(lambda: call('smd_conv3x3_thin_fwd', xp.data_ptr(), wt.data_ptr(), yt.data_ptr(), B, C, h, w, stream()))
            a_d = timeit(lambda: call('smd_conv3x3_thin_bwd', xp.data_ptr(), wt.data_ptr(), gy.data_ptr(), gxt.data_ptr(), None, None, 0, B, C, h, w, stream()))
            a_w = timeit(lambda: call('smd_conv3x3_thin_bwd', xp.data_ptr(), wt.data_ptr(), gy.data_ptr(), None, gwt.data_ptr(), wst.data_ptr(), nb, B, C, h, w, stream()))
            print(f'        f32MFMA fwd {a_f:6.1f} ({flop/a_f/1e6:5.1f} TF/s) data {a_d:7.1f} ({flop/a_d/1e6:5.1f}) wgt {a_w:7.1f} ({flop/a_w/1e6:5.1f})')
        line = f'        MIOpen fwd {t_f:7.1f} ({flop/t_f/1e6:5.1f} TF/s) data {t_d:7.1f} ({flop/t_d/1e6:5.1f}) wgt {t_w:7.1f} ({flop/t_w/1e6:5.1f})'
        for P in args.pieces:
            nb = _lib.lib.smd_conv3x3_mfma_packed_bytes(C, CO, P)
            wf = torch.empty(nb, device='cuda', dtype=torch.uint8); wb = torch.empty(nb, device='cuda', dtype=torch.uint8)
            y = torch.empty(B, CO, h, w, device='cuda'); gx = torch.empty_like(xp); gw = torch.empty_like(wt)
            nws = _lib.lib.smd_conv3x3_mfma_workspace_bytes(B, C, CO, h, w); ws = torch.empty(max(nws, 256), device='cuda', dtype=torch.uint8)
            t_p = timeit(lambda: call('smd_conv3x3_mfma_pack', wt.data_ptr(), wf.data_ptr(), wb.data_ptr(), C, CO, P, stream()))
            k_f = timeit(lambda: call('smd_conv3x3_mfma_fwd', xp.data_ptr(), wf.data_ptr(), y.data_ptr(), ws.data_ptr(), nws, B, C, CO, h, w, P, stream()))
            k_d = timeit(lambda: call('smd_conv3x3_mfma_bwd_data', gy.data_ptr(), wb.data_ptr(), gx.data_ptr(), ws.data_ptr(), nws, B, C, CO, h, w, P, stream()))
            k_w = timeit(lambda: call('smd_conv3x3_mfma_bwd_weight', xp.data_ptr(), gy.data_ptr(), gw.data_ptr(), ws.data_ptr(), nws, B, C, CO, h, w, P, stream()))
            line += f'\n        x{P}     fwd {k_f:7.1f} ({flop/k_f/1e6:5.1f} TF/s) data {k_d:7.1f} ({flop/k_d/1e6:5.1f}) wgt {k_w:7.1f} ({flop/k_w/1e6:5.1f})  pack {t_p:.1f}  partials {nws/1e6:.1f} MB'
        print(line, flush=True)
    except Exception:
        print(f'{name}: FAILED'); traceback.print_exc(); sys.stdout.flush()
