#!/usr/bin/env python3
"""Only one operator of the split-bf16 convolutions, for a kernel trace / counter pass (GPU box): conv_only.py fwd|data|wgt C CO h w [b] [pieces]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from slowtv_monodepth_amd import _lib
from slowtv_monodepth_amd._lib import call
op = sys.argv[1]; C, CO, h, w = map(int, sys.argv[2:6]); B = int(sys.argv[6]) if len(sys.argv) > 6 else 12; P = int(sys.argv[7]) if len(sys.argv) > 7 else 3
xp = torch.randn(B, C, h + 2, w + 2, device='cuda'); gy = torch.randn(B, CO, h, w, device='cuda'); wt = torch.randn(CO, C, 3, 3, device='cuda'); gw = torch.empty_like(wt)
y = torch.empty(B, CO, h, w, device='cuda'); gx = torch.empty_like(xp)
nb = _lib.lib.smd_conv3x3_mfma_packed_bytes(C, CO, P); wf = torch.empty(nb, device='cuda', dtype=torch.uint8); wb = torch.empty(nb, device='cuda', dtype=torch.uint8)
nws = _lib.lib.smd_conv3x3_mfma_workspace_bytes(B, C, CO, h, w); ws = torch.empty(max(nws, 256), device='cuda', dtype=torch.uint8)
st = torch.cuda.current_stream().cuda_stream
call('smd_conv3x3_mfma_pack', wt.data_ptr(), wf.data_ptr(), wb.data_ptr(), C, CO, P, st)
for _ in range(25):
    if op == 'fwd': call('smd_conv3x3_mfma_fwd', xp.data_ptr(), wf.data_ptr(), y.data_ptr(), ws.data_ptr(), nws, B, C, CO, h, w, P, st)
    elif op == 'data': call('smd_conv3x3_mfma_bwd_data', gy.data_ptr(), wb.data_ptr(), gx.data_ptr(), ws.data_ptr(), nws, B, C, CO, h, w, P, st)
    else: call('smd_conv3x3_mfma_bwd_weight', xp.data_ptr(), gy.data_ptr(), gw.data_ptr(), ws.data_ptr(), nws, B, C, CO, h, w, P, st)
torch.cuda.synchronize()
print('done')
