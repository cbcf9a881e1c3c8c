import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from slowtv_monodepth_amd import _lib
x = torch.randn(12, 16, 194, 642, device='cuda'); wt = torch.randn(16, 16, 3, 3, device='cuda'); y = torch.empty(12, 16, 192, 640, device='cuda')
st = torch.cuda.current_stream().cuda_stream
def run(): _lib.lib.smd_conv3x3_thin_fwd(x.data_ptr(), wt.data_ptr(), y.data_ptr(), 12, 16, 192, 640, st)
for _ in range(5): run()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(50): run()
e.record(); torch.cuda.synchronize()
print('THIN_MODE', os.environ.get('THIN_MODE'), f'{s.elapsed_time(e)/50*1e3:.1f} us')
