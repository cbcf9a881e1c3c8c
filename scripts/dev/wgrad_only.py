#!/usr/bin/env python3
"""Only the weight-gradient entry point of the split-bf16 convolutions, for a kernel trace (GPU box): wgrad_only.py C CO h w [b] [pieces]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from slowtv_monodepth_amd import _lib
from slowtv_monodepth_amd._lib import call
C, CO, h, w = map(int, sys.argv[1:5]); B = int(sys.argv[5]) if len(sys.argv) > 5 else 12; P = int(sys.argv[6]) if len(sys.argv) > 6 else 3
xp = torch.randn(B, C, h + 2, w + 2, device='cuda'); gy = torch.randn(B, CO, h, w, device='cuda'); gw = torch.empty(CO, C, 3, 3, device='cuda')
nws = _lib.lib.smd_conv3x3_mfma_workspace_bytes(B, C, CO, h, w); ws = torch.empty(max(nws, 256), device='cuda', dtype=torch.uint8)
st = torch.cuda.current_stream().cuda_stream
for _ in range(25): call('smd_conv3x3_mfma_bwd_weight', xp.data_ptr(), gy.data_ptr(), gw.data_ptr(), ws.data_ptr(), nws, B, C, CO, h, w, P, st)
torch.cuda.synchronize()
print('done', nws)
