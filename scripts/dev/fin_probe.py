import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from slowtv_monodepth_amd import functional as F
B, h, w = 12, 192, 640
g = torch.Generator(device='cuda').manual_seed(0)
inp = torch.rand(B, 3, h, w, device='cuda', generator=g)
depth = (1 + 10*torch.rand(B, 1, h, w, device='cuda', generator=g)).requires_grad_(True)
T = torch.eye(4, device='cuda').repeat(B, 1, 1); T[:, :3, 3] = 0.05; T.requires_grad_(True)
K = torch.tensor([[0.58*w, 0, 0.5*w, 0], [0, 1.92*h, 0.5*h, 0], [0, 0, 1, 0], [0, 0, 0, 1]], device='cuda')[None].repeat(B, 1, 1)
for _ in range(20):
    warp, dw, valid = F.view_synth(inp, depth, T, K)
    warp.sum().backward()
torch.cuda.synchronize()
