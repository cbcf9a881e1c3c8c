#!/usr/bin/env python3
"""Kernel-level timing of the fused forward / backward at a BASELINE shape, for tuning (run on the GPU box).
usage: microbench.py [cfg2|cfg4|cfg5] [iters]   — honours the SMD_* tuning env vars."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from slowtv_monodepth_amd import functional as F, _lib
from slowtv_monodepth_amd.synthetic import make_batch
name = sys.argv[1] if len(sys.argv) > 1 else 'cfg2'
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
b, h, w, supp = {'cfg2': (12, 192, 640, (-1, 1)), 'cfg4': (12, 384, 640, (-1, 1)), 'cfg5': (12, 384, 640, (-2, -1, 1, 2))}[name]
S, n = 4, len(supp)
dev = 'cuda'
_, y, _ = make_batch(b, h, w, supp, seed=42, device=dev)
g = torch.Generator(device=dev).manual_seed(0)
import torch.nn.functional as Fn
rough = os.environ.get('MB_ROUGH', '0') == '1'   # 1: per-pixel random disparity (incoherent gathers, worst case)
noise_amp = float(os.environ.get('MB_NOISE', '0.01'))   # per-pixel uniform noise on top of the smooth field (0: a converged network's output)
def mk(s):
    hs, ws = h >> s, w >> s
    if rough: return 0.05 + 0.9*torch.rand(b, 1, hs, ws, device=dev, generator=g)
    low = 0.2 + 0.6*torch.rand(b, 1, 4, 10, device=dev, generator=g)   # smooth field, like a network's output
    return Fn.interpolate(low, size=(hs, ws), mode='bilinear', align_corners=False) + noise_amp*torch.rand(b, 1, hs, ws, device=dev, generator=g)
disps = [mk(s).requires_grad_(True) for s in range(S)]
T = torch.eye(4, device=dev).repeat(n, b, 1, 1); T[..., :3, 3] = 0.05*torch.randn(n, b, 3, device=dev, generator=g); T.requires_grad_(True)
flags = F.recon_flags('ssim', True, True)
want_err = os.environ.get("MB_ERR", "0") == "1"   # 1: also write the per-pixel error map (an optional output)
fused_k0 = os.environ.get('MB_DISP', '1') == '1'   # 1: K0 fused into the reconstruction kernel (the product path); 0: separate K0 launch
prep_mode = os.environ.get('MB_PREP', 'inline')   # inline | ahead (prepared frames, consumed right away) | cold (prepared, then 1 GiB of unrelated traffic before the forward)
flush_src = torch.empty(1 << 28, device=dev) if prep_mode == 'cold' else None
one_node = os.environ.get('MB_PATH', 'handlers') == 'node'   # node: the single-node loss path (functional.loss_path_fused) with the pose leaves; handlers: the separate operators
for kv in filter(None, os.environ.get('MB_KNOBS', '').split(',')):   # e.g. MB_KNOBS=loss_path_guests=0,fwd_rh=12 (smd_set_knob)
    k_, v_ = kv.split('='); assert _lib.set_knob(k_, int(v_)), f'knob {k_} is not in this build'
aa = (0.01*torch.randn(n*b, 3, device=dev, generator=g)).requires_grad_(True); tt = (0.05*torch.randn(n*b, 3, device=dev, generator=g)).requires_grad_(True)
def step():
    prepared = None
    if prep_mode != 'inline':
        prepared = F.image_recon_prep(y['imgs'], y['supp_imgs'], flags=flags, pyramid=[d.shape[-2:] for d in disps] if fused_k0 else None, smooth_edges=fused_k0)
        if flush_src is not None: flush_src.add_(1.0)     # reads + writes 1 GiB: evicts L2 and the 256 MB Infinity Cache
    if one_node:
        Ts = F.pose_matrices(aa, tt).unflatten(0, (n, b))
        loss, *_ = F.loss_path_fused({s: d for s, d in enumerate(disps)}, y['imgs'], y['supp_imgs'], Ts, y['K'], pose=(aa, tt, None), flags=flags, min_depth=0.1, max_depth=100,
                                     seed=1, prepared=prepared)
        loss.backward()
        return loss
    if fused_k0: loss, err, sel, _, _ = F.image_recon_fused_disp(disps, y['imgs'], y['supp_imgs'], T, y['K'], flags=flags, min_depth=0.1, max_depth=100, seed=1, want_err=want_err, prepared=prepared)
    else:
        depth_up, _ = F.disp_to_depth(disps, (h, w), 0.1, 100)
        loss, err, sel, _ = F.image_recon_fused(depth_up, y["imgs"], y["supp_imgs"], T, y["K"], flags=flags, seed=1, want_err=want_err)
    lsm, _, _ = F.disp_smooth_fused({s: d for s, d in enumerate(disps)}, y['imgs'], use_edges=True, want_aux=False, prepared=prepared)
    (loss + 0.001*lsm).backward()
    return loss
for _ in range(3): step()
torch.cuda.synchronize()
for k in range(5): _lib.lib.smd_profile_enable(k, iters)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters): l = step()
e1.record(); torch.cuda.synchronize()
def col(which):
    buf = (C.c_float*iters)(); k = C.c_int(0)
    _lib.lib.smd_profile_collect(which, buf, iters, C.byref(k)); v = sorted(buf[i] for i in range(k.value))
    return v[len(v)//2]*1e3, v[0]*1e3
f, fb = col(0); bw, bb = col(1); fa, _ = col(2); ba, _ = col(3); pr, _ = col(4)
B = b*h*w*(S*9 + 12*(1 + n))
tag = ' '.join(f'{k}={v}' for k, v in os.environ.items() if k.startswith('SMD_') or k.startswith('MB_'))
print(f'{name} [{tag}] fwd med {f:.1f} us (min {fb:.1f}) = {B/f/1e3:.0f} GB/s | bwd med {bw:.1f} us (min {bb:.1f}) = {B/bw/1e3:.0f} GB/s | '
      f'entry points: fwd {fa:.1f} (of which prep {pr:.1f}) bwd {ba:.1f} us | whole loss path fwd+bwd {e0.elapsed_time(e1)/iters*1e3:.0f} us/iter | loss {l.item():.6f}')
