import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from slowtv_monodepth_amd import functional as F
b, h, w = 12, 192, 640
img = torch.rand(b, 3, h, w, device='cuda')
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/n*1e3
for scales in ([0], [1], [2], [3], [0, 1, 2, 3]):
    disps = {s: torch.rand(b, 1, h >> s, w >> s, device='cuda') for s in scales}
    for edges in (True, False):
        print(f'scales {scales} edges {edges}: fwd (main+finalize) {t(lambda: F.disp_smooth_fused(disps, img, use_edges=edges, want_aux=False)):.1f} us')
