#!/usr/bin/env python3
"""Where do the HIP and oracle gradients differ at a BASELINE shape?  (GPU box)  usage: dbg_fullsize_grad.py [b h w]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import torch
from oracle import view_synth_oracle as O
from slowtv_monodepth_amd import functional as F
from test_gpu_parity import _baseline_inputs
b, h, w = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (2, 192, 640)
supp, S = (-1, 1), 4
n = len(supp)
y, disps, aa, t, noise = _baseline_inputs(b, h, w, supp, S, seed=7)
def run(dev, hip):
    leaf = lambda v: v.detach().clone().to(dev).requires_grad_(True)
    d = {s: leaf(v) for s, v in disps.items()}
    a_, t_ = leaf(aa), leaf(t)
    imgs, sup = y['imgs'].to(dev), y['supp_imgs'].to(dev)
    if hip:
        Ts = F.pose_matrices(a_.flatten(0, 1), t_.flatten(0, 1)).unflatten(0, (n, b))
        depth_up, _ = F.disp_to_depth(list(d.values()), (h, w), 0.1, 100)
        depth_up.retain_grad()
        loss, err, sel, _ = F.image_recon_fused(depth_up, imgs, sup, Ts, y['K'].to(dev), flags=F.recon_flags('ssim', True, True), noise=noise.to(dev))
    else:
        Ts = O.T_from_AAt(a_.flatten(0, 1), t_.flatten(0, 1)).unflatten(0, (n, b))
        _, depth_up = O.disp_to_depth_up(d, (h, w), 0.1, 100, aten=True)
        for v in depth_up.values(): v.retain_grad()
        loss, ld, full = O.image_recon(depth_up, imgs, sup, Ts, y['K'], 'ssim', True, True, noise, True)
        sel = full['sel']
    loss.backward()
    gd = depth_up.grad.cpu().reshape(S, b, h, w) if hip else torch.stack([v.grad for v in depth_up.values()]).reshape(S, b, h, w)
    return loss.item(), sel.cpu().reshape(S, b, h, w), gd, a_.grad.cpu(), t_.grad.cpu()
lh, sh, gh, ah, th = run('cuda', True)
lr, sr, gr, ar, tr = run('cpu', False)
print('loss', lh, lr, 'flips', int((sh != sr).sum()))
d = (gh - gr).abs()
mx = gr.abs().max()
print('g_depth max ref', mx.item(), 'max diff', d.max().item(), 'rel', (d.max()/mx).item())
bad = d > 1e-3*mx
print('pixels with diff > 1e-3 max:', int(bad.sum()), 'of', bad.numel())
flipd = torch.nn.functional.max_pool2d((sh != sr).float().reshape(S*b, 1, h, w), 5, 1, 2).reshape(S, b, h, w) > 0
print('  of which within 2 px of a selection flip:', int((bad & flipd).sum()))
rest = bad & ~flipd
print('  elsewhere:', int(rest.sum()))
if rest.any():
    idx = rest.nonzero()
    print('  first few (s,b,v,u):', idx[:12].tolist())
    print('  column histogram (by 62):', torch.bincount(idx[:, 3]//62, minlength=11).tolist())
    print('  row histogram (by 12):', torch.bincount(idx[:, 2]//12, minlength=16).tolist())
    print('  by scale:', torch.bincount(idx[:, 0], minlength=S).tolist(), 'by sample:', torch.bincount(idx[:, 1], minlength=b).tolist())
print('aa rel', ((ah - ar).abs().max()/ar.abs().max()).item(), 't rel', ((th - tr).abs().max()/tr.abs().max()).item())
print('aa hip', ah.flatten()[:6].tolist(), '\naa ref', ar.flatten()[:6].tolist())
# ---- the largest differences, with what the oracle thinks about that pixel
order = d.flatten().argsort(descending=True)[:12]
for o in order.tolist():
    s_, b_, v_, u_ = (o // (b*h*w)), (o // (h*w)) % b, (o // w) % h, o % w
    print(f'  (s={s_}, b={b_}, v={v_}, u={u_}): hip {gh[s_, b_, v_, u_].item():+.3e} ref {gr[s_, b_, v_, u_].item():+.3e}  sel hip/ref '
          f'{sh[s_, b_, max(v_-1,0):v_+2, max(u_-1,0):u_+2].flatten().tolist()} / {sr[s_, b_, max(v_-1,0):v_+2, max(u_-1,0):u_+2].flatten().tolist()} '
          f'near flip: {bool(flipd[s_, b_, v_, u_])}')
