import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from slowtv_monodepth_amd.synthetic import make_batch
from slowtv_monodepth_amd.trainer import MonoDepthModule
wl = dict(bench.WORKLOADS['cfg2']); dev = torch.device('cuda:0')
m = MonoDepthModule(bench.make_cfg(wl, False)).to(dev)
x, y, _ = make_batch(12, 192, 640, wl['supp'], seed=1, device=dev)
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/n
def depth():
    out = m.nets['depth'](x['imgs']); sum(v.mean() for v in out['disp'].values()).backward()
def pose():
    pin = torch.cat([torch.cat([x['imgs'], s], 1) for s in x['supp_imgs']], 0)
    out = m.nets['pose'](pin); (out['R'].mean() + out['t'].mean()).backward()
def enc():
    f = m.nets['depth'].encoder(x['imgs']); sum(v.mean() for v in f).backward()
print(f'depth net fwd+bwd {t(depth):.2f} ms | depth encoder only {t(enc):.2f} ms | pose net fwd+bwd {t(pose):.2f} ms')
