#!/usr/bin/env python3
"""Where do two training steps differ between prep-ahead and inline prep (tests/test_gpu_scheduling.py)?  (GPU box)"""
import os, sys, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from slowtv_monodepth_amd import functional as Fm
from slowtv_monodepth_amd.synthetic import make_batch
from slowtv_monodepth_amd.trainer import MonoDepthModule

batches = [make_batch(12, 192, 640, (-1, 1), seed=42 + k, device='cuda') for k in range(2)]
real = Fm.image_recon_fused_disp
seen = []
def spy(*a, **kw):
    out = real(*a, **kw); seen.append((out[2], out[4])); return out
Fm.image_recon_fused_disp = spy

def run(prep_ahead, sync_between=False):
    seen.clear()
    torch.manual_seed(0)
    cfg = bench.make_cfg(bench.WORKLOADS['cfg2']); cfg['trainer']['prep_ahead'] = prep_ahead
    m = MonoDepthModule(cfg).cuda().train()
    res = []
    for batch in batches:
        for p in m.parameters(): p.grad = None
        loss, ld, fwd = m.step(batch)
        outs = {f'disp{s}': fwd['disp'][s] for s in sorted(fwd['disp'])}; outs['Ts'] = fwd['Ts']
        for o in outs.values(): o.retain_grad()
        loss.backward()
        if sync_between: torch.cuda.synchronize()
        res.append(dict(loss=loss.detach(), l_rec=ld['loss_img_recon'].detach(), l_sm=ld['loss_disp_smooth'].detach(), sel=seen[-1][0], depth=seen[-1][1].detach(),
                        outs={k: v.detach() for k, v in outs.items()}, grads={k: v.grad for k, v in outs.items()}))
    torch.cuda.synchronize()
    return res

def cmp(a, b, what):
    print(f'--- {what}')
    for k in range(2):
        line = [f'step {k}: loss {"==" if torch.equal(a[k]["loss"], b[k]["loss"]) else "!="} l_rec {"==" if torch.equal(a[k]["l_rec"], b[k]["l_rec"]) else "!="} '
                f'l_sm {"==" if torch.equal(a[k]["l_sm"], b[k]["l_sm"]) else "!="} sel diff {(a[k]["sel"] != b[k]["sel"]).sum().item()} depth diff {(a[k]["depth"] != b[k]["depth"]).sum().item()}']
        for name in a[k]['outs']:
            do = (a[k]['outs'][name] != b[k]['outs'][name]).sum().item()
            dg = (a[k]['grads'][name] != b[k]['grads'][name])
            line.append(f'{name}: out diff {do}, grad diff {dg.sum().item()} (max {(a[k]["grads"][name] - b[k]["grads"][name]).abs().max().item():.2e} of {a[k]["grads"][name].abs().max().item():.2e})')
            if dg.any() and name.startswith('disp'):
                idx = dg.nonzero()[:6].tolist(); line.append(f'   first at {idx}')
        print('\n   '.join(line))

i1 = run(False); i2 = run(False)
cmp(i1, i2, 'inline vs inline (same placement twice)')
a1 = run('pose'); a2 = run('pose')
cmp(a1, a2, 'prep-ahead vs prep-ahead')
cmp(a1, i1, 'prep-ahead vs inline')
s1 = run('pose', sync_between=True)
cmp(s1, i1, 'prep-ahead with a sync after every backward vs inline')
