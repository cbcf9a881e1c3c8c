import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from slowtv_monodepth_amd import functional as F
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/n*1e3
for (N, C, H, W) in ((12, 128, 96, 160), (12, 256, 48, 80), (12, 512, 24, 40), (12, 1024, 12, 20), (12, 96, 48, 160)):
    x = torch.randn(N, C, H, W, device='cuda', requires_grad=True); w = torch.randn(C, 1, 7, 7, device='cuda', requires_grad=True); b = torch.randn(C, device='cuda', requires_grad=True)
    y = F.dwconv7x7(x, w, b); g = torch.randn_like(y)
    tf = t(lambda: F.dwconv7x7(x, w, b))
    tb = t(lambda: torch.autograd.grad(y, (x, w, b), g, retain_graph=True))
    el = N*C*H*W
    print(f'N{N} C{C} {H}x{W}: fwd {tf:7.1f} us ({el*98/tf/1e6:6.2f} TFLOP/s, {el*8/tf/1e3:6.1f} GB/s)  bwd(data+weight) {tb:7.1f} us')
