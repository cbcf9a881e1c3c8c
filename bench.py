#!/usr/bin/env python3
"""Headline benchmark: training images/sec on synthetic 640x192 three-frame triplets + roofline of the fused kernel.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step = one full optimizer step of BASELINE.json configs[1] (ResNet-18 depth + ResNet-18 pose, b=12 per GPU, 640x192,
2 support frames, 4 scales, min-reprojection + automask + edge-aware smoothness, AdamW): network forward (PyTorch-ROCm),
K0 + fused reconstruction + smoothness (HIP, this repository), backward of both, optimizer.  Inputs are resident in HBM
before the timed region.  Rank 0 prints ONE JSON line (see DESIGN.md "Measurement" for every field).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path: sys.path.insert(0, str(ROOT))

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E datasheet peak (/opt/skills/guides/MI355X_MICROARCH.md)

WORKLOADS = {  # BASELINE.json configs[1..4]
    'cfg2': dict(depth='resnet18', pose='resnet18', h=192, w=640, b=12, supp=[-1, 1], learn_K=False, precision=32),
    'cfg3': dict(depth='convnext_tiny', pose='resnet18', h=192, w=640, b=12, supp=[-1, 1], learn_K=False, precision=32),
    'cfg4': dict(depth='convnext_tiny', pose='resnet18', h=384, w=640, b=12, supp=[-1, 1], learn_K=True, precision=32),
    'cfg5': dict(depth='convnext_base', pose='convnext_tiny', h=384, w=640, b=12, supp=[-2, -1, 1, 2], learn_K=True, precision='bf16'),
    'cfg1': dict(depth='resnet18', pose='resnet18', h=96, w=128, b=2, supp=[-1, 1], learn_K=False, precision=32),
}


def make_cfg(wl: dict, channels_last: bool = False) -> dict:
    return {
        'net': {'depth': {'enc_name': wl['depth'], 'pretrained': False, 'dec_name': 'monodepth', 'out_scales': [0, 1, 2, 3]},
                'pose': {'enc_name': wl['pose'], 'pretrained': False, 'learn_K': wl['learn_K']}},
        'loss': {'img_recon': {'weight': 1, 'loss_name': 'ssim', 'use_min': True, 'use_automask': True},
                 'disp_smooth': {'weight': 0.001, 'use_edges': True}},
        'optimizer': {'type': 'adamw', 'lr': 1e-4, 'weight_decay': 1e-3},
        'trainer': {'min_depth': 0.1, 'max_depth': 100, 'precision': wl['precision'], 'channels_last': channels_last,
                    'overlap_nets': os.environ.get('SMD_OVERLAP_NETS', '1') != '0'},
    }


def recon_bytes(b, h, w, n, S):
    """Algorithmic bytes of the fused forward / backward."""
    fwd = b*h*w*(S*(4 + 4 + 1) + 12*(1 + n))
    bwd = b*h*w*(S*(4 + 1 + 4) + 12*(1 + n))
    return fwd, bwd


def collect_profile(lib, which: int, cap: int):
    buf = (C.c_float*cap)(); n = C.c_int(0)
    lib.smd_profile_collect(which, buf, cap, C.byref(n))
    return [buf[i] for i in range(n.value)]


def cpu_baseline(wl: dict, sample_b: int = 4, steps: int = 5) -> dict:
    """The same training step on the host cores: PyTorch CPU networks + the CPU oracle as loss path (kind 'port')."""
    from oracle.backend import OracleBackend
    from slowtv_monodepth_amd.synthetic import make_batch
    from slowtv_monodepth_amd.train import StepModule, train_steps
    from slowtv_monodepth_amd.trainer import MonoDepthModule
    avail = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    cores = int(os.environ.get('SMD_CPU_THREADS', min(avail, 16)))   # measured on the 256-thread GPU host: 16 threads is the fastest setting (8: 3.1, 16: 4.7, 32: 3.3, 64: 1.8 img/s)
    torch.set_num_threads(cores)
    torch.manual_seed(42)
    module = MonoDepthModule(make_cfg({**wl, 'precision': 32}), loss_backend=OracleBackend(aten=True))
    opt = module.configure_optimizers()['optimizer']
    batch = make_batch(sample_b, wl['h'], wl['w'], wl['supp'], seed=42)
    model = StepModule(module)
    train_steps(model, opt, lambda it: batch, 1)
    t0 = time.perf_counter()
    train_steps(model, opt, lambda it: batch, steps)
    dt = (time.perf_counter() - t0)/steps
    return {'value': round(sample_b/dt, 3), 'unit': 'images/s', 'cores': cores, 'kind': 'port',
            'sample': f'{steps} full training steps (PyTorch-CPU nets + oracle loss path, fp32) on {sample_b} of the {wl["b"]} triplets '
                      f'of the workload, after 1 warm-up; {dt*1e3:.0f} ms/step'}


BASELINE_METRIC = 'training images/sec (640\u00d7192, 3-frame) at 1/2/4/8 MI355X; warp+SSIM HBM GB/s'   # BASELINE.json's metric, verbatim


def measured_hbm_ceilings(lib, device, nbytes=1 << 30, reps=10):
    """(copy GB/s counting read + write, read-only GB/s) of a STREAM-style sweep over `nbytes` through the library's own kernel."""
    src = torch.empty(nbytes, device=device, dtype=torch.uint8).fill_(1); dst = torch.empty_like(src)
    st = torch.cuda.current_stream().cuda_stream
    out = []
    for mode, factor in ((0, 2), (1, 1)):
        for _ in range(2): lib.smd_debug_stream_copy(src.data_ptr(), dst.data_ptr(), nbytes, mode, st)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps): lib.smd_debug_stream_copy(src.data_ptr(), dst.data_ptr(), nbytes, mode, st)
        e.record(); torch.cuda.synchronize()
        out.append(factor*nbytes*reps/(s.elapsed_time(e)*1e-3)/1e9)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--workload', default='cfg2', choices=sorted(WORKLOADS))
    ap.add_argument('--channels-last', action='store_true')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--precision', default=None, choices=['32', 'bf16'], help='override the network autocast precision of the workload (the loss path is always fp32)')
    args = ap.parse_args()

    from slowtv_monodepth_amd import _lib
    from slowtv_monodepth_amd.synthetic import make_batch
    from slowtv_monodepth_amd.train import StepModule, init_distributed, train_steps, wrap_ddp
    from slowtv_monodepth_amd.trainer import MonoDepthModule

    t_start = time.perf_counter()
    rank, local, world = init_distributed()
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run --nproc-per-node {args.gpus})'
    device = torch.device('cuda', local)
    torch.cuda.set_device(device)
    # NOTE: cudnn.benchmark (MIOpen exhaustive find) is left OFF: on a fresh box every candidate solver would be JIT-compiled
    # (tens of minutes); immediate mode compiles only the chosen kernel per convolution during the warm-up steps.
    wl = dict(WORKLOADS[args.workload])
    if args.precision: wl['precision'] = 32 if args.precision == '32' else 'bf16'
    torch.manual_seed(42)
    module = MonoDepthModule(make_cfg(wl, args.channels_last)).to(device)
    opt = module.configure_optimizers()['optimizer']
    batch = make_batch(wl['b'], wl['h'], wl['w'], wl['supp'], seed=42 + rank, device=device)
    model = wrap_ddp(StepModule(module), device)
    batch_fn = lambda it: batch

    def fence():
        if dist.is_initialized(): dist.barrier()
        torch.cuda.synchronize()

    def note(msg):
        if rank == 0: print(f'[bench +{time.perf_counter() - t_start:7.1f}s] {msg}', file=sys.stderr, flush=True)

    note('model built; warm-up (includes MIOpen kernel JIT on a fresh box)')
    for i in range(args.warmup):
        losses = train_steps(model, opt, batch_fn, 1)
        if i == 0: fence(); note('first step done')
    fence()
    note('warm-up done; timing')
    _lib.lib.smd_profile_enable(0, args.steps); _lib.lib.smd_profile_enable(1, args.steps)
    t0 = time.perf_counter()
    losses = train_steps(model, opt, batch_fn, args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    if dist.is_initialized():
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    fwd_ms = collect_profile(_lib.lib, 0, args.steps); bwd_ms = collect_profile(_lib.lib, 1, args.steps)
    _lib.lib.smd_profile_enable(0, 0); _lib.lib.smd_profile_enable(1, 0)
    last_loss = losses[-1].item()
    assert last_loss == last_loss, 'loss is NaN'

    if rank == 0:
        n, S = len(wl['supp']), 4
        B_fwd, B_bwd = recon_bytes(wl['b'], wl['h'], wl['w'], n, S)
        avg = lambda v: sum(v)/max(len(v), 1)
        f_ms, b_ms = avg(fwd_ms), avg(bwd_ms)
        copy_gbps, read_gbps = measured_hbm_ceilings(_lib.lib, device)
        traffic = None
        tf = ROOT/'profiles'/'traffic.json'   # per-launch HBM bytes from separate rocprofv3 --pmc passes (scripts/pmc_traffic.sh)
        if tf.is_file():
            try: traffic = json.loads(tf.read_text()).get(args.workload, {}).get('recon_fwd_bytes')
            except Exception: traffic = None
        out = {
            'metric': BASELINE_METRIC,
            'value': round(wl['b']*world*args.steps/elapsed, 2), 'unit': 'images/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(elapsed/args.steps*1e3, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'bf16' if wl['precision'] == 'bf16' else 'f32', 'data': 'synthetic',
            'config': {'workload': f'{args.workload}: {wl["depth"]} depth + {wl["pose"]} pose, {wl["w"]}x{wl["h"]}, {n} supports, 4 scales, '
                                   f'img_recon(ssim,min,automask)+disp_smooth(edges), AdamW, random init',
                       'global_batch': wl['b']*world, 'per_gpu_batch': wl['b'], 'parallelism': f'dp{world}',
                       'loss_dtype': 'f32', 'channels_last': args.channels_last, 'two_stream_nets': os.environ.get('SMD_OVERLAP_NETS', '1') != '0', 'final_loss': round(last_loss, 6)},
            'roofline': {'kernel': 'smd::k_recon_fwd<2,true> (fused warp+SSIM+L1+min-reproj+automask forward)', 'bound': 'hbm',
                         'achieved': round(B_fwd/(f_ms*1e-3)/1e9, 1) if f_ms else None, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                         'frac': round(B_fwd/(f_ms*1e-3)/1e9/HBM_PEAK_GBPS, 4) if f_ms else None, 'traffic': traffic,
                         'algorithmic_bytes': B_fwd, 'avg_kernel_ms': round(f_ms, 5), 'launches_timed': len(fwd_ms),
                         'peak_measured_copy': round(copy_gbps, 1), 'peak_measured_read': round(read_gbps, 1), 'frac_of_measured_copy': round(B_fwd/(f_ms*1e-3)/1e9/copy_gbps, 4) if f_ms else None},
            'roofline_bwd': {'kernel': 'smd::k_recon_bwd (fused adjoint)', 'bound': 'hbm',
                             'achieved': round(B_bwd/(b_ms*1e-3)/1e9, 1) if b_ms else None, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                             'frac': round(B_bwd/(b_ms*1e-3)/1e9/HBM_PEAK_GBPS, 4) if b_ms else None,
                             'algorithmic_bytes': B_bwd, 'avg_kernel_ms': round(b_ms, 5), 'launches_timed': len(bwd_ms)},
        }
        note(f'timed region done: {out["value"]} img/s')
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(wl)
            note('cpu baseline done')
        print(json.dumps(out), flush=True)
    if dist.is_initialized(): dist.destroy_process_group()


if __name__ == '__main__':
    main()
