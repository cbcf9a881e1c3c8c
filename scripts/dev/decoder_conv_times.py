#!/usr/bin/env python3
"""MIOpen's time for every convolution of the Monodepth decoder at cfg 2 (b = 12, 192x640, ResNet-18 skips), forward and backward, with the
FLOP/s and bytes/s they amount to: which layers a hand-written kernel could beat.  (GPU box.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from slowtv_monodepth_amd import miopen_tuning  # noqa: F401  (same find-db settings as the bench)
b = 12
layers = [('up0_4', 512, 256, 6, 20), ('up1_4', 512, 256, 12, 40), ('up0_3', 256, 128, 12, 40), ('up1_3', 256, 128, 24, 80), ('out_3', 128, 1, 24, 80),
          ('up0_2', 128, 64, 24, 80), ('up1_2', 128, 64, 48, 160), ('out_2', 64, 1, 48, 160), ('up0_1', 64, 32, 48, 160), ('up1_1', 96, 32, 96, 320),
          ('out_1', 32, 1, 96, 320), ('up0_0', 32, 16, 96, 320), ('up1_0', 16, 16, 192, 640), ('out_0', 16, 1, 192, 640)]
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/n*1e3
tf, tb = 0, 0
th = thb = tm = tmb = 0
from slowtv_monodepth_amd import functional as HF
print(f'{"layer":8s} {"cin":>4s} {"cout":>4s} {"h x w":>9s} {"fwd us":>8s} {"TF/s":>6s} {"GB/s":>6s} {"bwd us":>8s} {"TF/s":>6s}   (bwd = data + weight gradients)')
for name, ci, co, h, w in layers:
    x = torch.randn(b, ci, h + 2, w + 2, device='cuda', requires_grad=True)
    wt = torch.randn(co, ci, 3, 3, device='cuda', requires_grad=True)
    y = F.conv2d(x, wt); g = torch.randn_like(y)
    f = timeit(lambda: F.conv2d(x, wt))
    def bwd():
        x.grad = wt.grad = None
        y = F.conv2d(x, wt); y.backward(g)
    fb = timeit(bwd)
    flop = 2.0*b*h*w*ci*co*9
    byts = 4.0*b*(ci*(h + 2)*(w + 2) + co*h*w)
    tf += f; tb += fb - f
    print(f'{name:8s} {ci:4d} {co:4d} {h:4d}x{w:<4d} {f:8.1f} {flop/f/1e6:6.1f} {byts/f/1e3:6.0f} {fb - f:8.1f} {2*flop/(fb - f)/1e6:6.1f}')
    if co == 16:  # the library's direct convolution for the thin layers (smd_conv3x3_thin_*)
        hf = timeit(lambda: HF.conv3x3_thin(x, wt))
        def tbwd():
            x.grad = wt.grad = None
            HF.conv3x3_thin(x, wt).backward(g)
        hb = timeit(tbwd) - hf
        print(f'{"  thin":8s} {"":4s} {"":4s} {"":9s} {hf:8.1f} {flop/hf/1e6:6.1f} {byts/hf/1e3:6.0f} {hb:8.1f} {2*flop/hb/1e6:6.1f}   <- smd_conv3x3_thin')
    if co == 1:   # the library's stencil for one-channel heads (smd_conv3x3_head_*), sigmoid included
        bias = torch.zeros(1, device='cuda', requires_grad=True)
        hf = timeit(lambda: HF.conv3x3_head(x, wt, bias, 'sigmoid'))
        def hbwd():
            x.grad = wt.grad = bias.grad = None
            HF.conv3x3_head(x, wt, bias, 'sigmoid').backward(g)
        hb = timeit(hbwd) - hf
        th += hf; thb += hb; tm += f; tmb += fb - f
        print(f'{"  head":8s} {"":4s} {"":4s} {"":9s} {hf:8.1f} {flop/hf/1e6:6.1f} {byts/hf/1e3:6.0f} {hb:8.1f} {2*flop/hb/1e6:6.1f}   <- smd_conv3x3_head (with its sigmoid)')
print(f'total forward {tf:.0f} us, backward {tb:.0f} us')
print(f'the four heads: MIOpen {tm:.0f} + {tmb:.0f} us (without the sigmoid and its backward), stencil {th:.0f} + {thb:.0f} us')
