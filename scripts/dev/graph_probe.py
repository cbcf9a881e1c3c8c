#!/usr/bin/env python3
"""Which part of a training step survives HIP-graph capture?  Each stage runs in its own process (a failed capture may abort).  (GPU box)
usage: graph_probe.py            -> runs every stage as a subprocess
       graph_probe.py <stage>    -> one stage: loss | loss_prep | nets | opt | step_inline | step"""
import faulthandler, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
STAGES = ['loss', 'loss_prep', 'nets', 'opt', 'step_inline', 'step']
if len(sys.argv) < 2:
    for st in STAGES:
        r = subprocess.run([sys.executable, '-X', 'faulthandler', __file__, st], capture_output=True, text=True, timeout=600)
        tail = (r.stdout + r.stderr).strip().splitlines()
        keep = [l for l in tail if 'amdgpu.ids' not in l][-6:]
        print(f'=== {st}: rc={r.returncode}\n   ' + '\n   '.join(keep), flush=True)
    sys.exit(0)

faulthandler.enable()
import torch
import bench
from slowtv_monodepth_amd import functional as F, handlers
from slowtv_monodepth_amd.synthetic import make_batch
from slowtv_monodepth_amd.trainer import MonoDepthModule
import slowtv_monodepth_amd as amd
stage = sys.argv[1]
os.environ.setdefault('SMD_BWD_SKIP', '0')
dev = torch.device('cuda')
wl = bench.WORKLOADS['cfg2']
torch.manual_seed(0)
batch = make_batch(wl['b'], wl['h'], wl['w'], wl['supp'], seed=42, device=dev)
x, y, m = batch

def capture(fn, warm=2):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(warm): fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): out = fn()
    torch.cuda.synchronize()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    return out

if stage in ('loss', 'loss_prep'):
    b, h, w = wl['b'], wl['h'], wl['w']
    g = torch.Generator(device=dev).manual_seed(1)
    disps = [(0.05 + 0.9*torch.rand(b, 1, h >> s, w >> s, device=dev, generator=g)).requires_grad_(True) for s in range(4)]
    T = torch.eye(4, device=dev).repeat(2, b, 1, 1); T[..., :3, 3] = 0.05*torch.randn(2, b, 3, device=dev, generator=g); T.requires_grad_(True)
    crit = amd.losses.ReconstructionLoss(loss_name='ssim', use_min=True, use_automask=True); reg = amd.regularizers.SmoothReg(use_edges=True)
    def fn():
        for d in disps: d.grad = None
        T.grad = None
        prepared = None
        if stage == 'loss_prep':
            prepared = F.image_recon_prep(y['imgs'], y['supp_imgs'], flags=F.recon_flags('ssim', True, True), pyramid=[tuple(d.shape[-2:]) for d in disps],
                                          stream=torch.cuda.Stream(), smooth_edges=True)
        depths = handlers.LazyDepths([0, 1, 2, 3], disps, (h, w), 0.1, 100)
        l1, _ = handlers.image_recon(crit, None, depths, None, y['imgs'], y['supp_imgs'], T, y['K'], want_warp=False, prepared=prepared)
        l2, _ = handlers.disp_smooth(reg, dict(enumerate(disps)), y['imgs'], want_aux=False, prepared=prepared)
        loss = l1 + 0.001*l2
        loss.backward()
        return loss
    out = capture(fn)
    print('captured; loss', out.item())
else:
    cfg = bench.make_cfg(wl, capturable=True)
    if stage == 'step_inline': cfg['trainer']['prep_ahead'] = False
    module = MonoDepthModule(cfg).to(dev)
    opt = module.configure_optimizers()['optimizer']
    if stage == 'nets':
        def fn():
            for p in module.parameters(): p.grad = None
            fwd = module.forward(x)
            loss = sum(v.mean() for v in fwd['disp'].values()) + sum(v.mean() for k, v in fwd.items() if k.startswith('T_'))
            loss.backward()
            return loss
        out = capture(fn); print('captured; value', out.item())
    elif stage == 'opt':
        loss, _, _ = module.step(batch); loss.backward()
        out = capture(lambda: opt.step(), warm=1); print('captured optimizer step')
    else:
        def fn():
            loss, _, _ = module.step(batch)
            loss.backward()
            opt.step()
            return loss
        for p in module.parameters(): p.grad = None
        # (grads must not exist before capture: the captured backward then WRITES them)
        def first():
            opt.zero_grad(set_to_none=True); return fn()
        side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2): first()
        torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
        opt.zero_grad(set_to_none=True)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g): out = fn()
        torch.cuda.synchronize()
        import time
        for _ in range(3): g.replay()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): g.replay()
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f'captured whole step; loss {out.item():.6f}; replay: host {1e3*(t1 - t0)/10:.3f} ms/step, wall {1e3*(t2 - t0)/10:.3f} ms/step')
