#!/usr/bin/env python3
"""What would running the smoothness launches BESIDE the reconstruction kernels (second stream) buy?  (GPU box)

The wave traces (profiles/r04_fwd_wave_traces.txt) show the fused kernels draining for the last fifth of their span with one or two waves per
SIMD; the smoothness sweep / adjoint are small independent launches that could use that.  The loss path (prepared frames, K0-fused
reconstruction + smoothness, forward and backward) is captured into a HIP graph twice — everything on one stream, and the smoothness
forked onto a second stream (autograd runs its backward on that stream too) — and the replays are timed: GPU time of the loss path without
the host.  Gradients of the two variants are compared bit for bit.
usage: overlap_probe.py [cfg2|cfg4|cfg5] [replays]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as Fn
from slowtv_monodepth_amd import functional as F
from slowtv_monodepth_amd.synthetic import make_batch

name = sys.argv[1] if len(sys.argv) > 1 else 'cfg2'
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
os.environ.setdefault('SMD_BWD_SKIP', '0')
b, h, w, supp = {'cfg2': (12, 192, 640, (-1, 1)), 'cfg4': (12, 384, 640, (-1, 1)), 'cfg5': (12, 384, 640, (-2, -1, 1, 2))}[name]
S, n, dev = 4, len(supp), 'cuda'
_, y, _ = make_batch(b, h, w, supp, seed=42, device=dev)
g = torch.Generator(device=dev).manual_seed(0)
def mk(s):
    low = 0.2 + 0.6*torch.rand(b, 1, 4, 10, device=dev, generator=g)
    return Fn.interpolate(low, size=(h >> s, w >> s), mode='bilinear', align_corners=False) + 0.01*torch.rand(b, 1, h >> s, w >> s, device=dev, generator=g)
disps = [mk(s).requires_grad_(True) for s in range(S)]
T = torch.eye(4, device=dev).repeat(n, b, 1, 1); T[..., :3, 3] = 0.05*torch.randn(n, b, 3, device=dev, generator=g); T.requires_grad_(True)
flags = F.recon_flags('ssim', True, True)
prepared = F.image_recon_prep(y['imgs'], y['supp_imgs'], flags=flags, pyramid=[d.shape[-2:] for d in disps], smooth_edges=True)
torch.cuda.synchronize()
side = torch.cuda.Stream()

def step(fork):
    for d in disps: d.grad = None
    T.grad = None
    cur = torch.cuda.current_stream()
    if fork:
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            lsm, _, _ = F.disp_smooth_fused(dict(enumerate(disps)), y['imgs'], use_edges=True, want_aux=False, prepared=prepared)
    loss, _, sel, _, _ = F.image_recon_fused_disp(disps, y['imgs'], y['supp_imgs'], T, y['K'], flags=flags, min_depth=0.1, max_depth=100, seed=1, want_err=False, prepared=prepared)
    if fork: cur.wait_stream(side)
    else: lsm, _, _ = F.disp_smooth_fused(dict(enumerate(disps)), y['imgs'], use_edges=True, want_aux=False, prepared=prepared)
    tot = loss + 0.001*lsm
    tot.backward()
    return tot

res = {}
for fork in (False, True):
    warm = torch.cuda.Stream(); warm.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(warm):
        for _ in range(3): step(fork)
    torch.cuda.current_stream().wait_stream(warm); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, capture_error_mode='thread_local'): out = step(fork)
    torch.cuda.synchronize()
    for _ in range(5): gr.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps):
        e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1)*1e3)
    ts.sort()
    res[fork] = (ts[len(ts)//2], ts[0], out.detach().clone(), [d.grad.clone() for d in disps] + [T.grad.clone()])
    print(f'{name}: smoothness {"on a second stream" if fork else "after the reconstruction, one stream"}: loss path (forward + backward) {ts[len(ts)//2]:.1f} us per replay (min {ts[0]:.1f}); loss {out.item():.8f}', flush=True)
same = torch.equal(res[False][2], res[True][2]) and all(torch.equal(a, c) for a, c in zip(res[False][3], res[True][3]))
print(f'{name}: gained {res[False][0] - res[True][0]:.1f} us per step; loss and every gradient bit-identical: {same}')
