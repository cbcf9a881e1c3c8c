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


def make_cfg(wl: dict, channels_last: bool = False, capturable: bool = False) -> dict:
    return {
        'net': {'depth': {'enc_name': wl['depth'], 'pretrained': False, 'dec_name': 'monodepth', 'out_scales': [0, 1, 2, 3]},
                'pose': {'enc_name': wl['pose'], 'pretrained': False, 'learn_K': wl['learn_K']}},
        'loss': {'img_recon': {'weight': 1, 'loss_name': 'ssim', 'use_min': True, 'use_automask': True},
                 'disp_smooth': {'weight': 0.001, 'use_edges': True}},
        'optimizer': {'type': 'adamw', 'lr': 1e-4, 'weight_decay': 1e-3, **({'capturable': True} if capturable else {})},
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


def cpu_model() -> str:
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'): return line.split(':', 1)[1].strip()
    except OSError: pass
    import platform
    return platform.processor() or 'unknown'


def cpu_baseline(wl: dict, sample_b: int = 2, steps: int = 10, warmup: int = 3) -> dict:
    """the same work on the host cores, median of `steps` after `warmup` — (1) the loss path alone (oracle,
    forward + backward from given network outputs), (2) the full training step (PyTorch-CPU networks + the oracle loss path,
    kind 'port').  Bounded sample: `sample_b` of the workload's triplets, so the default run stays within a few minutes."""
    import statistics
    from oracle import view_synth_oracle as O
    from oracle.backend import OracleBackend
    from slowtv_monodepth_amd.synthetic import make_batch
    from slowtv_monodepth_amd.train import StepModule, train_steps
    from slowtv_monodepth_amd.trainer import MonoDepthModule
    avail = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    cores = int(os.environ.get('SMD_CPU_THREADS', min(avail, 16)))   # measured on the 256-thread GPU host: 16 threads is the fastest setting (8: 3.1, 16: 4.7, 32: 3.3, 64: 1.8 img/s)
    torch.set_num_threads(cores)
    torch.manual_seed(42)
    h, w, n, S = wl['h'], wl['w'], len(wl['supp']), 4
    batch = make_batch(sample_b, h, w, wl['supp'], seed=42)
    # (1) loss path only
    g = torch.Generator().manual_seed(0)
    disps = {s: (0.05 + 0.9*torch.rand(sample_b, 1, h >> s, w >> s, generator=g)).requires_grad_(True) for s in range(S)}
    Ts = torch.eye(4).repeat(n, sample_b, 1, 1); Ts[..., :3, 3] = 0.05*torch.randn(n, sample_b, 3, generator=g); Ts.requires_grad_(True)
    y = batch[1]

    def loss_step():
        for v in disps.values(): v.grad = None
        Ts.grad = None
        t0 = time.perf_counter()
        loss, _ = O.loss_path(disps, y['imgs'], y['supp_imgs'], Ts, y['K'], aten=True)
        loss.backward()
        return time.perf_counter() - t0
    for _ in range(warmup): loss_step()
    t_loss = statistics.median(loss_step() for _ in range(steps))
    B_fwd, B_bwd = recon_bytes(sample_b, h, w, n, S)
    # (2) full training step
    module = MonoDepthModule(make_cfg({**wl, 'precision': 32}), loss_backend=OracleBackend(aten=True))
    opt = module.configure_optimizers()['optimizer']
    model = StepModule(module)
    train_steps(model, opt, lambda it: batch, warmup)
    ts = []
    for _ in range(steps):
        t0 = time.perf_counter()
        train_steps(model, opt, lambda it: batch, 1)
        ts.append(time.perf_counter() - t0)
    dt = statistics.median(ts)
    return {'value': round(sample_b/dt, 3), 'unit': 'images/s', 'cores': cores, 'kind': 'port', 'cpu_model': cpu_model(), 'host_threads_available': avail,
            'loss_path_only': {'value': round(sample_b/t_loss, 3), 'unit': 'images/s', 'ms_per_step': round(t_loss*1e3, 1),
                               'effective_GBps': round((B_fwd + B_bwd)/t_loss/1e9, 3)},
            'sample': f'median of {steps} steps after {warmup} warm-ups on {sample_b} of the {wl["b"]} triplets of the workload, fp32: '
                      f'loss path only (oracle forward + backward, {t_loss*1e3:.0f} ms) and full training step (PyTorch-CPU nets + oracle '
                      f'loss path, {dt*1e3:.0f} ms)'}


BASELINE_METRIC = 'training images/sec (640\u00d7192, 3-frame) at 1/2/4/8 MI355X; warp+SSIM HBM GB/s'   # BASELINE.json's metric, verbatim


def measured_hbm_ceilings(lib, device, nbytes=1 << 30, reps=10):
    """(copy GB/s counting read + write, read-only GB/s) of a STREAM-style sweep over `nbytes` through the library's own kernel."""
    src = torch.empty(nbytes, device=device, dtype=torch.uint8).fill_(1); dst = torch.empty_like(src)
    st = torch.cuda.current_stream().cuda_stream
    out = [0.0, 0.0]
    for mode, factor in ((0, 2), (1, 1), (2, 2), (3, 1), (4, 2), (5, 1)):   # 2, 3: deeper unroll; 4, 5: block-contiguous chunks; quote the best
        for _ in range(2): lib.smd_debug_stream_copy(src.data_ptr(), dst.data_ptr(), nbytes, mode, st)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps): lib.smd_debug_stream_copy(src.data_ptr(), dst.data_ptr(), nbytes, mode, st)
        e.record(); torch.cuda.synchronize()
        out[mode & 1] = max(out[mode & 1], factor*nbytes*reps/(s.elapsed_time(e)*1e-3)/1e9)
    return out


def decoder_roofline(wl, device):
    """The decoder's own kernels on the workload's full-resolution shapes, outside the timed region (driver-timed, not only under
    profiles/): HIP events on the launch stream around 20 raw C calls each.  The thin last stage `conv3x3(16 -> 16)` in both of its forms — the f32 MFMA
    (smd_conv3x3_thin_*: exact f32 products at the vector rate, peak 157.3 TFLOP/s) and the bf16 matrix cores with three-way split operands
    (smd_conv3x3_mfma_*: six bf16 products per f32 product; bound by HBM once the arithmetic costs 6/16) — as fp32-equivalent TFLOP/s against the f32
    matrix peak and as bytes against 8 TB/s; the widest-image wide layer (32 output channels at half resolution); the full-resolution head as a stencil."""
    from slowtv_monodepth_amd import _lib, functional as F
    call, lib = _lib.call, _lib.lib
    B, h, w = wl['b'], wl['h'], wl['w']
    st = lambda: torch.cuda.current_stream().cuda_stream
    def timeit(fn, n=20):
        for _ in range(3): fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n): fn()
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e)/n
    gen = torch.Generator(device=device).manual_seed(7)
    rnd = lambda *sh: torch.randn(*sh, device=device, generator=gen)
    out = {'note': 'HIP events (torch.cuda.Event on the launch stream) over 20 raw C calls per operator, after the timed region; TFLOP/s = fp32-equivalent 2 x 9 x C x CO x pixels / time; '
                   'f32_mfma_peak 157.3 TFLOP/s (v_mfma_f32_*_f32 = the vector rate), hbm_peak 8000 GB/s; bytes = 4 (C (h+2)(w+2) + CO h w) b'}
    def conv_block(C, CO, hh, ww):
        xp, wt, gy = rnd(B, C, hh + 2, ww + 2), rnd(CO, C, 3, 3)/(3*C**0.5), rnd(B, CO, hh, ww)
        y, gx, gw = torch.empty(B, CO, hh, ww, device=device), torch.empty_like(xp), torch.empty_like(wt)
        flop, byts = 2.0*9*C*CO*B*hh*ww, 4.0*B*(C*(hh + 2)*(ww + 2) + CO*hh*ww)
        nb = lib.smd_conv3x3_mfma_packed_bytes(C, CO, 3)
        wf, wb = torch.empty(nb, device=device, dtype=torch.uint8), torch.empty(nb, device=device, dtype=torch.uint8)
        nws = lib.smd_conv3x3_mfma_workspace_bytes(B, C, CO, hh, ww); ws = torch.empty(max(nws, 256), device=device, dtype=torch.uint8)
        call('smd_conv3x3_mfma_pack', wt.data_ptr(), wf.data_ptr(), wb.data_ptr(), C, CO, 3, st())
        ms = {'fwd': timeit(lambda: call('smd_conv3x3_mfma_fwd', xp.data_ptr(), wf.data_ptr(), y.data_ptr(), ws.data_ptr(), nws, B, C, CO, hh, ww, 3, st())),
              'bwd_data': timeit(lambda: call('smd_conv3x3_mfma_bwd_data', gy.data_ptr(), wb.data_ptr(), gx.data_ptr(), ws.data_ptr(), nws, B, C, CO, hh, ww, 3, st())),
              'bwd_weight': timeit(lambda: call('smd_conv3x3_mfma_bwd_weight', xp.data_ptr(), gy.data_ptr(), gw.data_ptr(), ws.data_ptr(), nws, B, C, CO, hh, ww, 3, st()))}
        blk = {'shape': f'{C}->{CO} at {hh}x{ww}, b={B}', 'split_bf16_mfma': {k: {'ms': round(v, 5), 'tflops': round(flop/v/1e9, 1), 'frac_of_f32_mfma_peak': round(flop/v/1e9/157.3, 3),
                                                                                  'GBps': round(byts/v/1e6, 1), 'frac_of_hbm_peak': round(byts/v/1e6/HBM_PEAK_GBPS, 3)} for k, v in ms.items()}}
        if CO == 16:
            nt = lib.smd_conv3x3_thin_workspace_bytes(B, C, hh, ww); wt_ws = torch.empty(max(nt, 256), device=device, dtype=torch.uint8)
            mt = {'fwd': timeit(lambda: call('smd_conv3x3_thin_fwd', xp.data_ptr(), wt.data_ptr(), y.data_ptr(), B, C, hh, ww, st())),
                  'bwd_data': timeit(lambda: call('smd_conv3x3_thin_bwd', xp.data_ptr(), wt.data_ptr(), gy.data_ptr(), gx.data_ptr(), None, None, 0, B, C, hh, ww, st())),
                  'bwd_weight': timeit(lambda: call('smd_conv3x3_thin_bwd', xp.data_ptr(), wt.data_ptr(), gy.data_ptr(), None, gw.data_ptr(), wt_ws.data_ptr(), nt, B, C, hh, ww, st()))}
            blk['f32_mfma'] = {k: {'ms': round(v, 5), 'tflops': round(flop/v/1e9, 1), 'frac_of_f32_mfma_peak': round(flop/v/1e9/157.3, 3)} for k, v in mt.items()}
        return blk
    out['thin_stage'] = conv_block(16, 16, h, w)
    out['thin_stage']['bound'] = 'mfma (f32 form) / hbm (split-bf16 form)'
    out['wide_stage'] = conv_block(96 if wl['depth'].startswith('resnet') else 32, 32, h//2, w//2)
    out['wide_stage']['bound'] = 'mfma (bf16 pipe: 6 products per f32 product; ceiling 2.67 x the f32 matrix peak)'
    xp, wt1, bs = rnd(B, 16, h + 2, w + 2), rnd(1, 16, 3, 3)/12, rnd(1)
    y1 = torch.empty(B, 1, h, w, device=device)
    t_h = timeit(lambda: call('smd_conv3x3_head_fwd', xp.data_ptr(), wt1.data_ptr(), bs.data_ptr(), y1.data_ptr(), B, 16, h, w, 1, st()))
    hb = 4.0*B*(16*(h + 2)*(w + 2) + h*w)
    out['head_full_resolution_fwd'] = {'shape': f'16->1 at {h}x{w}, b={B} (+ sigmoid)', 'bound': 'hbm', 'ms': round(t_h, 5), 'GBps': round(hb/t_h/1e6, 1), 'frac_of_hbm_peak': round(hb/t_h/1e6/HBM_PEAK_GBPS, 3)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--workload', default='cfg2', choices=sorted(WORKLOADS))
    ap.add_argument('--channels-last', action='store_true')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--graph', action='store_true', help='capture one whole training step (networks, loss path, backward, optimizer) into a HIP graph after the warm-up and time its '
                                                          'replays: what the host has to enqueue per step drops from ~1000 launches to one (multi-GPU readiness: eight ranks\' Python '
                                                          'threads); the library\'s event profile is off in this mode, so the roofline fields are null')
    ap.add_argument('--knob', action='append', default=[], metavar='NAME=VALUE', help='pin a launch-shape knob of the library (smd_set_knob) for an A/B run, e.g. --knob bwd_live=0; recorded in config.knobs')
    ap.add_argument('--conv-route', default='auto', choices=['auto', 'mfma', 'miopen'], help="who serves the decoder's wide convolutions: 'auto' = this box's A/B per operator and shape on first use "
                    "(functional._conv_route; recorded in config.decoder_conv_routes), 'mfma' / 'miopen' pin every one of them for an A/B run")
    ap.add_argument('--precision', default=None, choices=['32', 'bf16'], help='override the network autocast precision of the workload (the loss path is always fp32)')
    args = ap.parse_args()

    from slowtv_monodepth_amd import _lib
    from slowtv_monodepth_amd.synthetic import make_batch
    from slowtv_monodepth_amd.train import StepModule, init_distributed, train_steps, wrap_ddp
    from slowtv_monodepth_amd.trainer import MonoDepthModule

    from slowtv_monodepth_amd import functional as _HF
    _HF.set_conv_route(args.conv_route)
    for kv in args.knob:
        k_, v_ = kv.split('=')
        if not _lib.set_knob(k_, int(v_)): raise SystemExit(f'bench.py: knob {k_} is not in this build of the library')
    t_start = time.perf_counter()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # not under a launcher: start one rank per GPU ourselves (one process per GPU, RCCL process group)
        import socket, subprocess
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0)); port = sk.getsockname()[1]
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1',
               '--master-port', str(port), str(Path(__file__).resolve())] + sys.argv[1:]
        sys.exit(subprocess.run(cmd, env={**os.environ, 'HSA_ENABLE_IPC_MODE_LEGACY': os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0')}).returncode)
    rank, local, world = init_distributed()
    if world != args.gpus: raise SystemExit(f'bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks')
    device = torch.device('cuda', local)
    torch.cuda.set_device(device)
    # NOTE: cudnn.benchmark (MIOpen exhaustive find) is left OFF: on a fresh box every candidate solver would be JIT-compiled
    # (tens of minutes); immediate mode compiles only the chosen kernel per convolution during the warm-up steps.
    wl = dict(WORKLOADS[args.workload])
    if args.precision: wl['precision'] = 32 if args.precision == '32' else 'bf16'
    torch.manual_seed(42)
    cfg = make_cfg(wl, args.channels_last, capturable=args.graph)
    # Under capture the frame-only prep runs inline (on the capturing stream): with the prep enqueued on the pose network's side stream
    # `hipStreamEndCapture` crashes on this stack (scripts/dev/graph_probe.py: every piece of the step captures, the whole step with inline prep
    # captures, only that cross-stream hand-off of a buffer allocated inside the capture does not).  Costs the graph 36 + 16 us of a 16.7 ms step.
    if args.graph: cfg['trainer']['prep_ahead'] = False
    module = MonoDepthModule(cfg).to(device)
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
    graph_note = None
    run_steps = lambda k: train_steps(model, opt, batch_fn, k)
    if args.graph:
        # One training step as ONE HIP graph.  The row loop of the fused backward is whatever the tuner chose during the warm-up (it does not
        # time inside a capture); nothing in the step reads the host: inputs, the in-kernel tie-break seed and the optimizer's step counter
        # (capturable AdamW) live on the device.  NOTE: the tie-break seed of `ReconstructionLoss` is a launch argument, so every replay
        # draws the SAME tie-break noise (it only decides exact ties of the automask).
        from slowtv_monodepth_amd.train import FlatAllReduce
        try:
            side = torch.cuda.Stream(device)
            side.wait_stream(torch.cuda.current_stream(device))
            with torch.cuda.stream(side):                     # PyTorch's capture recipe: a few eager steps on a side stream first
                for _ in range(2): train_steps(model, opt, batch_fn, 1)
            torch.cuda.current_stream(device).wait_stream(side)
            fence()
            opt.zero_grad(set_to_none=True)
            graph = torch.cuda.CUDAGraph()
            # thread_local: with a process group, RCCL's watchdog thread polls the events of the warm-up steps' collectives; under the default
            # (global) mode such a query from ANOTHER thread invalidates the capture and the watchdog dies with it
            if isinstance(model, FlatAllReduce):
                # data-parallel: forward + backward are one graph, the optimizer step a second one, and the gradient all-reduces run eagerly
                # between the two (`FlatAllReduce.average_static`): no collective inside a capture
                model.require_sync = False                    # the wrapper's gradient hooks stay inert, now and on every replay
                with torch.cuda.graph(graph, capture_error_mode='thread_local'):
                    loss_static, _ = model(batch)
                    loss_static.backward()
                grads = [[p.grad if p.grad is not None else torch.zeros_like(p) for p in bk] for bk in model.buckets]   # rewritten in place by every replay
                for bk, views in zip(model.buckets, model.views):
                    for p, v in zip(bk, views): p.grad = v    # what the optimizer reads: its slice of the averaged bucket
                model.average_static(grads)
                graph_opt = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph_opt, capture_error_mode='thread_local'): opt.step()
                def run_steps(k):
                    for _ in range(k): graph.replay(); model.average_static(grads); graph_opt.replay()
                    return [loss_static.detach()]
                what = 'forward of both networks on two streams, loss path and backward as one graph, the optimizer step as a second one, the gradient all-reduces eagerly between them'
            else:
                with torch.cuda.graph(graph, capture_error_mode='thread_local'):
                    loss_static, _ = model(batch)
                    loss_static.backward()
                    opt.step()
                def run_steps(k):
                    for _ in range(k): graph.replay()
                    return [loss_static.detach()]
                what = 'forward of both networks on two streams, loss path, backward, optimizer step'
            run_steps(2); fence()
            launch_ms = []
            for _ in range(5):                                # what ONE step costs the host when the queue is empty (the timed loop below runs into the
                t_l = time.perf_counter(); run_steps(1)       # runtime's back-pressure: its enqueue time tends to the GPU time)
                launch_ms.append((time.perf_counter() - t_l)*1e3); fence()
            graph_note = f'whole step captured ({what}) and replayed; one step costs the host {sorted(launch_ms)[2]:.3f} ms (median of 5, queue empty)'
        except Exception as e:   # e.g. a collective that cannot be captured: report it, time the eager loop
            graph_note = f'capture failed ({type(e).__name__}: {str(e)[:200]}); eager loop timed instead'
            torch.cuda.synchronize()
            run_steps = lambda k: train_steps(model, opt, batch_fn, k)
        note(graph_note)
    note('warm-up done; timing')
    profiled = not (args.graph and graph_note and graph_note.startswith('whole step'))
    if profiled:
        for which in range(5): _lib.lib.smd_profile_enable(which, args.steps)
    t0, c0 = time.perf_counter(), time.process_time()
    losses = run_steps(args.steps)
    host_enqueue = time.perf_counter() - t0      # nothing in the loop synchronises: this is the Python / ATen front end's time to ENQUEUE the steps
    host_cpu = time.process_time() - c0          # ... and the CPU time of this process over the same span (all threads: a blocked launch call does not count)
    fence()
    elapsed = time.perf_counter() - t0
    if dist.is_initialized():
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    kernel_variants = (_lib.lib.smd_last_kernel_variant(0).decode() or 'unknown', _lib.lib.smd_last_kernel_variant(1).decode() or 'unknown')
    fwd_ms = collect_profile(_lib.lib, 0, args.steps); bwd_ms = collect_profile(_lib.lib, 1, args.steps)
    fwd_all_ms = collect_profile(_lib.lib, 2, args.steps); bwd_all_ms = collect_profile(_lib.lib, 3, args.steps)
    prep_ms = collect_profile(_lib.lib, 4, args.steps)
    for which in range(5): _lib.lib.smd_profile_enable(which, 0)
    # The same kernel with the frame-only prep run INLINE, right before it (packed buffer warm in the caches), outside the timed region:
    # prep-ahead takes 36 us off the critical path but hands the kernel a buffer that went cold under the networks.
    fwd_inline_ms = []
    if profiled and rank == 0 and not dist.is_initialized() and not args.no_cpu_baseline:   # (a profiler's steady-state window is the LAST steps of the process: the profile scripts pass --no-cpu-baseline)
        ahead = module.prep_ahead
        module.prep_ahead = False
        train_steps(model, opt, batch_fn, 2)
        _lib.lib.smd_profile_enable(0, 8)
        train_steps(model, opt, batch_fn, 8); torch.cuda.synchronize()
        fwd_inline_ms = collect_profile(_lib.lib, 0, 8)
        _lib.lib.smd_profile_enable(0, 0)
        module.prep_ahead = ahead
    rccl_ranks = dist.get_world_size() if dist.is_initialized() else 1
    last_loss = losses[-1].item()
    assert last_loss == last_loss, 'loss is NaN'

    if rank == 0:
        n, S = len(wl['supp']), 4
        B_fwd, B_bwd = recon_bytes(wl['b'], wl['h'], wl['w'], n, S)
        avg = lambda v: sum(v)/max(len(v), 1)
        f_ms, b_ms, fa_ms, ba_ms, p_ms = avg(fwd_ms), avg(bwd_ms), avg(fwd_all_ms), avg(bwd_all_ms), avg(prep_ms)
        copy_gbps, read_gbps = measured_hbm_ceilings(_lib.lib, device)
        # HBM traffic cannot be measured inside this run (it needs rocprofv3 --pmc passes, which serialise the kernels): it is
        # read from the committed summary of such passes over THIS command (scripts/pmc_traffic.sh -> profiles/traffic.json).
        traffic, traffic_source, tj = None, None, {}
        tf = ROOT/'profiles'/'traffic.json'
        if tf.is_file():
            try:
                tj = json.loads(tf.read_text())
                traffic = tj.get(args.workload, {}).get('recon_fwd_bytes')
                if traffic: traffic_source = f"profiles/traffic.json ({tj.get('_source', 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes')})"
            except Exception: traffic = None
        from slowtv_monodepth_amd import functional as _F
        tuner = _F.row_skip_tuner(torch.cuda.current_device())
        skipping = (int(os.environ['SMD_BWD_SKIP']) >= 1) if 'SMD_BWD_SKIP' in os.environ else tuner.skip
        k_fwd, k_bwd = kernel_variants
        # What the selection maps of the LAST step looked like (outside the timed region): the share of auto-masked pixels and of pixels routed to each
        # support.  The backward's cost depends on it (a support nobody selects in a strip costs its wave nothing: the liveness table) — a bench whose
        # masks have collapsed (cfg 4 with randomly initialised learned intrinsics: > 99 % auto-masked from the second step on) times the all-masked floor.
        sel_stats = None
        sel = getattr(module.backend, 'last_sel', None)
        if sel is not None:
            sel_stats = {'automasked_share': round((sel == 255).float().mean().item(), 4),
                         'routed_share_per_support': [round((sel == i).float().mean().item(), 4) for i in range(n)],
                         'dead_wave_share_per_support': {'as_the_liveness_table_sees_it': [round(v, 4) for v in _F.dead_wave_shares(sel, True, n, table_rh=16).tolist()],
                                                         'exact_footprint': [round(v, 4) for v in _F.dead_wave_shares(sel, True, n).tolist()]}}
        out = {
            'metric': BASELINE_METRIC,
            'value': round(wl['b']*world*args.steps/elapsed, 2), 'unit': 'images/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(elapsed/args.steps*1e3, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'bf16' if wl['precision'] == 'bf16' else 'f32', 'data': 'synthetic',
            'config': {'workload': f'{args.workload}: {wl["depth"]} depth + {wl["pose"]} pose, {wl["w"]}x{wl["h"]}, {n} supports, 4 scales, '
                                   f'img_recon(ssim,min,automask)+disp_smooth(edges), AdamW, random init',
                       'global_batch': wl['b']*world, 'per_gpu_batch': wl['b'], 'parallelism': f'dp{world}',
                       'loss_dtype': 'f32', 'channels_last': args.channels_last, 'two_stream_nets': os.environ.get('SMD_OVERLAP_NETS', '1') != '0', 'final_loss': round(last_loss, 6),
                       'rccl_ranks': rccl_ranks, 'dp_impl': (os.environ.get('SMD_DP_IMPL', 'flat') if rccl_ranks > 1 or os.environ.get('SMD_FORCE_DDP') == '1' else None),
                       'host_enqueue_ms_per_step': round(host_enqueue/args.steps*1e3, 3), 'host_cpu_ms_per_step': round(host_cpu/args.steps*1e3, 3), 'hip_graph': graph_note,
                       'prep_ahead': module.prep_ahead if not args.graph else 'False (forced by --graph: the prep-ahead hand-off across streams crashes hipStreamEndCapture on this stack; the default bench runs prep_ahead=pose)',
                       'loss_path': getattr(module.backend, 'last_path', None), 'knobs': args.knob or None,
                       'decoder_conv_routes': {'mode': args.conv_route, 'decisions': {f'{k[0]} {k[2]}->{k[3]} {k[4]}x{k[5]} b{k[1]}': (('bf16 mfma' if k[0].endswith('_bf16') else 'split-bf16 mfma') if v[0] else ('f32 mfma' if (k[3] == 16 and not k[0].endswith('_bf16')) else 'miopen')) + f' ({v[1]:.0f} vs {v[2]:.0f} us)' for k, v in sorted(_HF.conv_routes().items(), key=lambda kv: (-kv[0][4], kv[0][0]))}},
                       'automasked_share': sel_stats['automasked_share'] if sel_stats else None, 'routed_share_per_support': sel_stats['routed_share_per_support'] if sel_stats else None,
                       'dead_wave_share_per_support': sel_stats['dead_wave_share_per_support'] if sel_stats else None},
            'roofline': {'kernel': f'{k_fwd} (disp->depth + warp + SSIM + L1 + min-reproj + automask forward in one launch; the instantiation the library reports for the last forward launch, name as rocprofv3 prints it)', 'bound': 'hbm',
                         'achieved': round(B_fwd/(f_ms*1e-3)/1e9, 1) if f_ms else None, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                         'frac': round(B_fwd/(f_ms*1e-3)/1e9/HBM_PEAK_GBPS, 4) if f_ms else None, 'traffic': traffic, 'traffic_source': traffic_source,
                         'whole_forward_ms': round(fa_ms, 5), 'whole_forward_launches': 'the forward entry point on the critical path: k_recon_main (all 4 scales; the loss is reduced inside it by the last block; in the single-node loss path the smoothness sweep rides in the same launch as guest blocks and the weighted sum is formed in-launch)',
                         'prep_ms': round(p_ms, 5), 'prep_launch': 'k_recon_prep (frame-only: texel repack, target window sums, identity error), enqueued on the pose network\'s side stream behind that network, i.e. under the depth network (trainer.prep_ahead)',
                         'whole_forward_frac': round(B_fwd/(fa_ms*1e-3)/1e9/HBM_PEAK_GBPS, 4) if fa_ms else None,
                         'forward_incl_prep_frac': round(B_fwd/((fa_ms + p_ms)*1e-3)/1e9/HBM_PEAK_GBPS, 4) if fa_ms else None,
                         'algorithmic_bytes': B_fwd, 'avg_kernel_ms': round(f_ms, 5), 'launches_timed': len(fwd_ms),
                         'avg_kernel_ms_inline_prep': round(avg(fwd_inline_ms), 5) if fwd_inline_ms else None,
                         'frac_inline_prep': round(B_fwd/(avg(fwd_inline_ms)*1e-3)/1e9/HBM_PEAK_GBPS, 4) if fwd_inline_ms else None,
                         # since round 5 the launch also carries the smoothness sweep as guest blocks (single-node loss path): `frac` stays on the reconstruction's
                         # bytes alone (comparable with earlier rounds and with BASELINE's "fused warp+SSIM+min-reproj kernel"); this is the launch's whole algorithmic work
                         'frac_incl_guest_smoothness_sweep': (round((B_fwd + wl['b']*wl['h']*wl['w']*17.3125)/(f_ms*1e-3)/1e9/HBM_PEAK_GBPS, 4)
                                                              if (f_ms and str(getattr(module.backend, 'last_path', '')).startswith('single node')) else None),
                         'guest_note': 'the forward launch also runs the smoothness sweep of handlers.disp_smooth (25.5 MB algorithmic at cfg 2: b*h*w*17.3125, SURVEY.md §8d) as guest blocks behind its own; its duration includes them',
                         'peak_measured_copy': round(copy_gbps, 1), 'peak_measured_read': round(read_gbps, 1), 'frac_of_measured_copy': round(B_fwd/(f_ms*1e-3)/1e9/copy_gbps, 4) if f_ms else None},
            'roofline_bwd': {'kernel': f'{k_bwd} (fused adjoint, one wave per (strip, support); the pose / intrinsics epilogue rides in the K0-adjoint launch that follows; the instantiation the library reports for the last backward launch)', 'bound': 'hbm',
                             'row_loop': {'dead_row_skipping': skipping, 'timed': tuner.last,
                                          'chosen_by': 'SMD_BWD_SKIP' if 'SMD_BWD_SKIP' in os.environ else 'functional.row_skip_tuner: four early backward calls of every period (16 calls, doubling up to 256 while the timings confirm the choice) alternate between the two row loops (same gradients bit for bit) with HIP events around the entry point; skipping is kept if it is more than 3 % faster (profiles/r04_skip_regimes.txt)',
                                          'liveness_table': 'on: a backward wave whose (strip, support) no pixel selects parks zeros instead of running its row loop (the forward records, per strip and support, the columns with such a pixel)' if not any(k.startswith('bwd_live=0') for k in args.knob) else 'off (--knob bwd_live=0)'},
                             'achieved': round(B_bwd/(b_ms*1e-3)/1e9, 1) if b_ms else None, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                             'frac': round(B_bwd/(b_ms*1e-3)/1e9/HBM_PEAK_GBPS, 4) if b_ms else None,
                             'algorithmic_bytes': B_bwd, 'avg_kernel_ms': round(b_ms, 5), 'launches_timed': len(bwd_ms),
                             'traffic': (tj.get(args.workload, {}).get('recon_bwd_bytes') if tf.is_file() else None),
                             'whole_backward_ms': round(ba_ms, 5), 'whole_backward_launches': 'k_recon_bwd alone (events around the fused backward\'s launch; the K0 adjoint — two launches, the first also carrying the pose epilogue through to the pose network\'s outputs and the smoothness adjoint as guest blocks — follows outside this pair)'},
        }
        note(f'timed region done: {out["value"]} img/s')
        if world == 1 and profiled:
            try: out['roofline_decoder'] = decoder_roofline(wl, device)
            except Exception as e: out['roofline_decoder'] = {'error': f'{type(e).__name__}: {str(e)[:200]}'}
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(wl)
            note('cpu baseline done')
        print(json.dumps(out), flush=True)
    if dist.is_initialized(): dist.destroy_process_group()


if __name__ == '__main__':
    main()
