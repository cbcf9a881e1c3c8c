#!/usr/bin/env python3
"""Phase timeline of the forward / data-gradient convolution kernel (GPU box; library built with EXTRA=-DSMD_CONV_TRACE, SMD_HOTPATH_LIB pointing at it):
shader-clock stamps of wave 0 of every block: start, after the prologue's staging, after its barrier; per chunk: start, after the taps, after filing the next
patch, after the barrier; end (after the stores).  usage: conv_trace.py C CO h w [fwd|data] [b]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from slowtv_monodepth_amd import _lib
from slowtv_monodepth_amd._lib import call
Cc, CO, h, w = map(int, sys.argv[1:5]); op = sys.argv[5] if len(sys.argv) > 5 else 'fwd'; B = int(sys.argv[6]) if len(sys.argv) > 6 else 12
xp = torch.randn(B, Cc, h + 2, w + 2, device='cuda'); wt = torch.randn(CO, Cc, 3, 3, device='cuda'); gy = torch.randn(B, CO, h, w, device='cuda')
nb = _lib.lib.smd_conv3x3_mfma_packed_bytes(Cc, CO, 3); wf = torch.empty(nb, device='cuda', dtype=torch.uint8); wb = torch.empty(nb, device='cuda', dtype=torch.uint8)
y = torch.empty(B, CO, h, w, device='cuda'); gx = torch.empty_like(xp)
nws = _lib.lib.smd_conv3x3_mfma_workspace_bytes(B, Cc, CO, h, w); ws = torch.empty(max(nws, 256), device='cuda', dtype=torch.uint8)
st = torch.cuda.current_stream().cuda_stream
call('smd_conv3x3_mfma_pack', wt.data_ptr(), wf.data_ptr(), wb.data_ptr(), Cc, CO, 3, st)
for _ in range(3):
    if op == 'fwd': call('smd_conv3x3_mfma_fwd', xp.data_ptr(), wf.data_ptr(), y.data_ptr(), ws.data_ptr(), nws, B, Cc, CO, h, w, 3, st)
    else: call('smd_conv3x3_mfma_bwd_data', gy.data_ptr(), wb.data_ptr(), gx.data_ptr(), ws.data_ptr(), nws, B, Cc, CO, h, w, 3, st)
torch.cuda.synchronize()
fn = _lib.lib.smd_debug_conv_trace; fn.restype = C.c_int
n = 8192
buf = np.zeros((n, 40), dtype=np.uint64)
assert fn(buf.ctypes.data_as(C.c_void_p), n) == 0
live = buf[:, 39] > 0
t = buf[live].astype(np.int64)
t0 = t[:, 0].min()
nch = int(((t[:, 3:39:4] > 0).sum(axis=1)).max())
clk = 100e6   # s_memtime on gfx9 counts the constant 100 MHz reference; printed in us
us = lambda x: x/clk*1e6
print('raw span ticks', int(t[:, 39].max() - t0)); print(f'{op} {Cc}->{CO} {h}x{w} b={B}: blocks traced {live.sum()}, chunks per block {nch}, launch span {us(t[:, 39].max() - t0):.1f} us')
print(f'block life: mean {us((t[:, 39] - t[:, 0]).mean()):.2f} us (p10 {us(np.percentile(t[:, 39] - t[:, 0], 10)):.2f}, p90 {us(np.percentile(t[:, 39] - t[:, 0], 90)):.2f})')
print(f'prologue (request + 4 fetches + file): {us((t[:, 1] - t[:, 0]).mean()):.2f} us, its barrier {us((t[:, 2] - t[:, 1]).mean()):.2f}')
for c in range(nch):
    s = 3 + 4*c
    ok = t[:, s] > 0
    print(f'chunk {c}: taps {us((t[ok, s + 1] - t[ok, s]).mean()):.2f} us | file {us((t[ok, s + 2] - t[ok, s + 1]).mean()):.2f} | barrier {us((t[ok, s + 3] - t[ok, s + 2]).mean()):.2f}')
last = 3 + 4*(nch - 1) + 3
print(f'epilogue (stores): {us((t[:, 39] - t[:, last]).mean()):.2f} us')
starts = np.sort(t[:, 0] - t0)
print('block starts (us) at 10 % steps:', ' '.join(f'{us(np.percentile(starts, p)):.1f}' for p in range(0, 101, 10)))
