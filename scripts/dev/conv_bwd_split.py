#!/usr/bin/env python3
"""MIOpen's data-gradient and weight-gradient times, separately, for the decoder's thin layers at cfg 2 (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from slowtv_monodepth_amd import miopen_tuning  # noqa: F401
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/n*1e3
for name, ci, co, h, w in [('up1_0', 16, 16, 192, 640), ('up0_0', 32, 16, 96, 320)]:
    x = torch.randn(12, ci, h + 2, w + 2, device='cuda'); wt = torch.randn(co, ci, 3, 3, device='cuda'); g = torch.randn(12, co, h, w, device='cuda')
    bw = lambda mask: torch.ops.aten.convolution_backward(g, x, wt, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, mask)
    print(name, f'fwd {timeit(lambda: F.conv2d(x, wt)):.1f} us, data gradient {timeit(lambda: bw([True, False, False])):.1f} us, weight gradient {timeit(lambda: bw([False, True, False])):.1f} us')
