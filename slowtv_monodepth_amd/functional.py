"""`torch.autograd.Function`s around the C ABI (`include/smd_hotpath.h`).

PyTorch's role here is plumbing only: it owns the device buffers, provides the current HIP stream and carries
the hand-written backward kernels in its autograd graph.  Every function validates its inputs on the host and
raises the exception types the reference raises (`ValueError`) before any launch; there is no fallback path —
CPU tensors are rejected.
"""
from __future__ import annotations

import os
import threading

import torch

from . import _lib
from ._lib import FLAGS, REGR_FLAGS, SEL_MASKED, int_array, ptr_array
from ._lib import call as _raw_call

__all__ = ['conv3x3_mfma', 'conv3x3_wide', 'set_conv_route', 'conv_routes', 'loss_path_fused', 'crop_resize', 'disp_to_depth', 'image_recon_prep', 'PreparedFrames', 'image_recon_fused', 'image_recon_fused_disp', 'disp_smooth_fused', 'view_synth', 'photo_error', 'recon_reduce',
           'lane_shift_selftest', 'recon_flags', 'regression_loss', 'elu_pad', 'elu_up_cat_pad', 'batch_norm_act', 'max_pool3x3s2', 'dwconv7x7', 'layer_norm_cf', 'pose_matrices', 'intrinsics', 'inv_intrinsics']


# Device of the operands of the operator this THREAD is executing: launches go to ITS current stream.  Thread-local, and set at
# the top of every forward (by `_check`) AND every backward (by `_on`): autograd runs the backward of each device on its own
# thread and has already made that device current there, so a process-wide "last validated device" would send the backward of
# one GPU's graph to another GPU's stream as soon as two devices are used in one process.
_tls = threading.local()


def _on(t: torch.Tensor) -> torch.device:
    """Declare `t`'s device the device of the operator being executed on this thread (call first in every backward)."""
    _tls.device = t.device
    return t.device


def call(name: str, *args):
    """Launch with the operands' device current (the library launches on the calling thread's current HIP device)."""
    dev = getattr(_tls, 'device', None)
    if dev is not None and dev.index is not None and dev.index != torch.cuda.current_device():
        with torch.cuda.device(dev): return _raw_call(name, *args)
    return _raw_call(name, *args)


def _stream() -> int:
    """The HIP stream of the operands' device.  (Not simply `torch.cuda.current_stream()`: with tensors on a GPU that is not the
    process's current device that would be a stream of another device.)"""
    dev = getattr(_tls, 'device', None)
    return torch.cuda.current_stream(dev if dev is not None else torch.cuda.current_device()).cuda_stream


def _check(name: str, t: torch.Tensor, shape=None) -> torch.Tensor:
    if not isinstance(t, torch.Tensor): raise TypeError(f'{name} must be a Tensor, got {type(t)}')
    if not t.is_cuda: raise RuntimeError(f'{name} must live on the GPU: the view-synthesis hot path has no CPU implementation')
    _tls.device = t.device
    if t.dtype != torch.float32: raise TypeError(f'{name} must be float32 (the loss path is fp32 only), got {t.dtype}')
    if shape is not None and tuple(t.shape) != tuple(shape): raise ValueError(f'{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}')
    return t.contiguous()


def recon_flags(loss_name: str = 'ssim', use_min: bool = False, use_automask: bool = False) -> int:
    if loss_name not in ('ssim', 'l1'): raise NotImplementedError(f"fused image reconstruction supports loss_name 'ssim'|'l1', not {loss_name!r}")
    return (FLAGS['use_min'] if use_min else 0) | (FLAGS['use_automask'] if use_automask else 0) | (FLAGS['loss_l1'] if loss_name == 'l1' else 0)


# ---------------------------------------------------------------------------------------------------
class _DispToDepth(torch.autograd.Function):
    """K0: per-scale bilinear upsample + `to_scaled`/`to_inv` (src/core/trainer.py:316-321)."""

    @staticmethod
    def forward(ctx, size, min_depth, max_depth, want_disp_up, *disps):
        h, w = size
        b = disps[0].shape[0]
        disps = [_check(f'disp[{i}]', d) for i, d in enumerate(disps)]
        for d in disps:
            if d.ndim != 4 or d.shape[0] != b or d.shape[1] != 1: raise ValueError(f'disparities must be (b,1,hs,ws), got {tuple(d.shape)}')
        S = len(disps)
        hs, ws = [d.shape[2] for d in disps], [d.shape[3] for d in disps]
        depth_up = torch.empty((S, b, 1, h, w), device=disps[0].device, dtype=torch.float32)
        disp_up = torch.empty_like(depth_up) if want_disp_up else None
        call('smd_disp_to_depth_fwd', ptr_array([d.data_ptr() for d in disps]), int_array(hs), int_array(ws), S, b, h, w,
             float(min_depth or 0), float(max_depth or 0), depth_up.data_ptr(), disp_up.data_ptr() if want_disp_up else None, _stream())
        ctx.save_for_backward(depth_up)
        ctx.meta = (hs, ws, S, b, h, w, float(min_depth or 0), float(max_depth or 0))
        if want_disp_up:
            ctx.mark_non_differentiable(disp_up)
            return depth_up, disp_up
        return depth_up, None

    @staticmethod
    def backward(ctx, g_depth_up, _g_disp_up):
        (depth_up,) = ctx.saved_tensors
        _on(depth_up)
        hs, ws, S, b, h, w, mn, mx = ctx.meta
        g_depth_up = _check('grad(depth_up)', g_depth_up)
        g_disps = [torch.empty((b, 1, hs[s], ws[s]), device=depth_up.device, dtype=torch.float32) for s in range(S)]
        hs_a, ws_a = int_array(hs), int_array(ws)
        nbytes = _lib.lib.smd_disp_to_depth_workspace_bytes(hs_a, ws_a, S, b, h, w)
        wsp = torch.empty(nbytes, device=depth_up.device, dtype=torch.uint8)
        call('smd_disp_to_depth_bwd', hs_a, ws_a, S, b, h, w, mn, mx, depth_up.data_ptr(), g_depth_up.data_ptr(),
             ptr_array([g.data_ptr() for g in g_disps]), wsp.data_ptr(), nbytes, _stream())
        return (None, None, None, None, *g_disps)


def disp_to_depth(disps, size, min_depth=None, max_depth=None, want_disp_up=False):
    """disps: sequence of (b,1,hs,ws) -> depth_up (S,b,1,h,w) [, disp_up (S,b,1,h,w)] in one launch."""
    if min_depth is not None and min_depth <= 0: raise ValueError(f'Min depth must be greater than 0. ({min_depth})')
    if max_depth and min_depth and max_depth < min_depth: raise ValueError(f'Max depth must be greater than min. ({max_depth} vs. {min_depth})')
    return _DispToDepth.apply(tuple(int(x) for x in size), min_depth, max_depth, bool(want_disp_up), *disps)


# ---------------------------------------------------------------------------------------------------
class PreparedFrames:
    """What the loss path needs from the FRAMES alone: the packed texel / target-window buffer of the reconstruction forward
    (`smd_image_recon_prep`), optionally the edge weights of the smoothness term for the same pyramid (`smd_disp_smooth_prep`), the HIP
    event after which they are complete, and what they were built for.  None of it depends on a network output, so the training step
    fills it on a side stream while the networks run (`MonoDepthModule.step`)."""
    def __init__(self, packed, event, key, edge_w=None):
        self.packed, self.event, self.key, self.edge_w = packed, event, key, edge_w

    def edges_for(self, imgs, hs, ws):
        """The edge-weight buffer if it was built for this frame and pyramid, else None."""
        if self.edge_w is None or self.key[0] != imgs.data_ptr() or self.key[2] != tuple(imgs.shape): return None
        return self.edge_w if (self.key[5] == tuple(hs) and self.key[6] == tuple(ws)) else None

    def matches(self, imgs, supp_imgs, flags, hs, ws) -> bool:
        k = (imgs.data_ptr(), supp_imgs.data_ptr(), tuple(imgs.shape), tuple(supp_imgs.shape), int(flags) & _PREP_FLAGS,
             tuple(hs) if hs is not None else None, tuple(ws) if ws is not None else None)
        return k == self.key


_PREP_FLAGS = FLAGS['use_min'] | FLAGS['use_automask'] | FLAGS['loss_l1']


def image_recon_prep(imgs, supp_imgs, *, flags: int, pyramid=None, stream=None, smooth_edges: bool = False) -> PreparedFrames:
    """Fill the frame-only buffer of the fused reconstruction for (imgs (b,3,h,w), supp_imgs (n,b,3,h,w)).

    :param flags: `recon_flags(...)` of the criterion that will consume it (the identity error of the automask is part of it).
    :param pyramid: [(hs, ws), ...] of the disparity pyramid when the K0-fused forward follows (its row table is built here).
    :param stream: `torch.cuda.Stream` to run on (default: the current one).  The returned object carries the completion event;
        the forward that consumes it waits for that event on ITS stream.
    :param smooth_edges: also compute the edge weights of `SmoothReg(use_edges=True)` for `pyramid` (`disp_smooth_fused(prepared=...)`)."""
    b, _, h, w = imgs.shape
    n = supp_imgs.shape[0]
    imgs_c = _check('imgs', imgs, (b, 3, h, w)); supp_c = _check('supp_imgs', supp_imgs, (n, b, 3, h, w))
    dev = imgs.device
    cur = torch.cuda.current_stream(dev)
    st = stream if stream is not None else cur
    hs = [int(p[0]) for p in pyramid] if pyramid else None
    ws = [int(p[1]) for p in pyramid] if pyramid else None
    if st is not cur: st.wait_stream(cur)          # the frames were produced on the caller's stream
    with torch.cuda.stream(st):
        packed = torch.empty(_lib.lib.smd_packed_supports_bytes(b, n, h, w)//4, device=dev, dtype=torch.float32)
        call('smd_image_recon_prep', imgs_c.data_ptr(), supp_c.data_ptr(), packed.data_ptr(), int_array(hs) if hs else None, int_array(ws) if ws else None,
             len(hs) if hs else 0, b, n, h, w, int(flags) & _PREP_FLAGS, st.cuda_stream)
        edge_w = None
        if smooth_edges and hs:
            hs_a, ws_a = int_array(hs), int_array(ws)
            edge_w = torch.empty(_lib.lib.smd_disp_smooth_edge_weight_bytes(hs_a, ws_a, len(hs), b), device=dev, dtype=torch.uint8)
            call('smd_disp_smooth_prep', imgs_c.data_ptr(), hs_a, ws_a, len(hs), b, h, w, FLAGS['use_edges'], edge_w.data_ptr(), st.cuda_stream)
        event = torch.cuda.Event()
        event.record(st)
    if st is not cur:
        for t in (imgs_c, supp_c): t.record_stream(st)
        packed.record_stream(cur)
        if edge_w is not None: edge_w.record_stream(cur)
    key = (imgs.data_ptr(), supp_imgs.data_ptr(), tuple(imgs.shape), tuple(supp_imgs.shape), int(flags) & _PREP_FLAGS,
           tuple(hs) if hs else None, tuple(ws) if ws else None)
    return PreparedFrames(packed, event, key, edge_w)


def _packed_for(prepared, imgs, supp, flags, hs, ws, b, n, h, w, dev):
    """-> (packed buffer, flags): the prepared one (after waiting for it on the current stream) or a fresh one for an inline prep."""
    if prepared is not None:
        if not prepared.matches(imgs, supp, flags, hs, ws):
            raise ValueError('PreparedFrames were built for other frames, flags or another disparity pyramid')
        cur = torch.cuda.current_stream(dev)
        cur.wait_event(prepared.event)
        prepared.packed.record_stream(cur)      # allocated on the stream that filled it, used (and later freed) on this one
        return prepared.packed, int(flags) | FLAGS['packed_ready']
    return torch.empty(_lib.lib.smd_packed_supports_bytes(b, n, h, w)//4, device=dev, dtype=torch.float32), int(flags)


def _need_err(want_err: bool, n: int) -> bool:
    return bool(want_err) or n > _lib.lib.smd_image_recon_supports_per_pass()


def dead_tile_shares(sel: torch.Tensor, use_min: bool, n: int = 1, cols: int = 60) -> torch.Tensor:
    """(n, S): per support and scale, the share of (image row, `cols`-column tile) units in which NO pixel routes gradient to that
    support — the rows the fused backward's dead-row skipping (`k_recon_bwd<…, SKIP=2>`) passes over in that support's wave.
    sel: (S,b,1,h,w)|(S,b,h,w) uint8."""
    S = sel.shape[0]
    s4 = sel.reshape(S, -1, sel.shape[-2], sel.shape[-1])
    pad = (-s4.shape[-1]) % cols
    def dead(live):                                  # -> (S,)
        live = torch.nn.functional.pad(live, (0, pad))
        return 1.0 - live.view(S, live.shape[1], live.shape[2], -1, cols).any(-1).float().mean(dim=(1, 2, 3))
    if not use_min: return dead(s4 != SEL_MASKED)[None].expand(max(int(n), 1), S)   # the mean over the supports: all live wherever the automask is not
    return torch.stack([dead(s4 == i) for i in range(max(int(n), 1))])


def dead_wave_shares(sel: torch.Tensor, use_min: bool, n: int = 1, rh: int = 16, cols: int = 60, table_rh: int | None = None) -> torch.Tensor:
    """(n,): per support, the share of the fused backward's (strip of `rh` rows x `cols` columns, support) waves in whose footprint — the strip
    dilated by one pixel: rows r0-1 .. r1, columns c0-1 .. c0+cols — NO pixel routes gradient to that support: the waves that park zeros instead
    of running their row loop.  `table_rh`: rows are rounded out to whole forward strips of that many rows, which is what the forward's
    liveness table resolves (smd_kernels.h); None: the exact footprint.  sel: (S,b,1,h,w)|(S,b,h,w) uint8 (host-side diagnostic, used by bench.py)."""
    S = sel.shape[0]
    s4 = sel.reshape(S, -1, sel.shape[-2], sel.shape[-1])
    h, w = s4.shape[-2:]
    out = []
    for k in range(max(int(n), 1)):
        live = (s4 == k) if use_min else (s4 != SEL_MASKED)
        dead, tot = 0, 0
        for r0 in range(0, h, rh):
            lo, hi = max(r0 - 1, 0), min(r0 + rh, h - 1)
            if table_rh: lo, hi = (lo//table_rh)*table_rh, min((hi//table_rh + 1)*table_rh - 1, h - 1)
            rows = live[:, :, lo:hi + 1].any(2)                       # (S,B,w)
            for c0 in range(0, w, cols):
                d = ~rows[:, :, max(c0 - 1, 0): min(c0 + cols, w - 1) + 1].any(2)
                dead += int(d.sum()); tot += d.numel()
        out.append(dead/max(tot, 1))
    return torch.tensor(out)


class _RowSkipTuner:
    """Chooses between the two row loops of the fused backward by timing them on the live data, without ever stalling the stream.

    The backward gives the same gradients bit for bit with or without dead-row skipping (`SMD_BWD_SKIP_DEAD_ROWS`); which is faster
    depends on the selection masks and on the geometry (`profiles/r04_skip_regimes.txt`): the plain loop is 15-24 % faster where every
    row of a wave's window has a live pixel (the masks of a training run at 192x640 from the second step on: 110 vs 135 us), the
    gated loop wins once 75-80 % of the (row, 60-column window) units are dead and takes less than half the time when the automask
    takes everything (52 vs 117 us; 384x640 with randomly initialised learned intrinsics: 110-120 vs 247 us) — and the share of
    masked pixels alone does not predict the sign.  So it is measured: after `settle` calls, `2*trials` backward calls of every period alternate between the two
    loops with a pair of HIP events around the entry point; later calls harvest the pairs that have completed (`Event.query`, no
    wait), and skipping is used from then on if its fastest trial beats the plain loop's by more than 3 %.  The period between two timings adapts (below).  `SMD_BWD_SKIP` in the
    environment pins the choice (a profiler perturbs the timing: `scripts/round_profiles.sh` pins what the un-traced run chose; pin it
    as well when capturing the step into a HIP graph — timing events cannot be recorded during capture)."""
    # The re-timing period adapts (round 5): the masks of a young network change within a few optimiser steps (profiles/r04_mask_runs.txt), those of
    # a trained one hardly at all — a period starts at `period_min` calls, doubles each time the timing confirms the previous choice (up to
    # `period_max`) and falls back to `period_min` when the choice flips.
    period_min, period_max, settle, trials, margin = 16, 256, 1, 2, 0.97

    def __init__(self):
        self.calls, self.skip, self.pending, self.samples, self.last = 0, False, [], {True: [], False: []}, None
        self.period, self.decided = self.period_min, 0

    def _flag(self, skip: bool) -> int:
        return FLAGS['bwd_skip_rows'] if skip else 0

    def _harvest(self) -> None:
        still = []
        for mode, e0, e1 in self.pending:
            if e1.query(): self.samples[mode].append(e0.elapsed_time(e1))
            else: still.append((mode, e0, e1))
        self.pending = still
        if len(self.samples[True]) >= self.trials and len(self.samples[False]) >= self.trials:
            t_skip, t_plain = min(self.samples[True]), min(self.samples[False])
            choice = t_skip < self.margin*t_plain
            self.period = min(2*self.period, self.period_max) if (self.decided and choice == self.skip) else self.period_min
            self.skip, self.decided = choice, self.decided + 1
            self.last = {'skipping_ms': round(t_skip, 5), 'plain_ms': round(t_plain, 5), 'next_period': self.period}
            self.samples = {True: [], False: []}

    def begin(self, dev):
        """-> (flag bits for this backward call, token for `end`)."""
        if 'SMD_BWD_SKIP' in os.environ: return self._flag(os.environ['SMD_BWD_SKIP'] not in ('', '0')), None   # pinned: no timing (read here, per call; the library itself never reads the environment)
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing(): return self._flag(self.skip), None   # no timing events inside a HIP-graph capture: the choice made so far is what gets captured
        if self.pending: self._harvest()
        if self.calls >= self.period: self.calls = 0       # a new period (its length may have changed at the last harvest)
        phase = self.calls - self.settle                   # (the first calls of a process carry one-off costs)
        self.calls += 1
        if 0 <= phase < 2*self.trials:
            mode = phase % 2 == 0
            e0 = torch.cuda.Event(enable_timing=True); e0.record(torch.cuda.current_stream(dev))
            return self._flag(mode), (mode, e0, dev)
        return self._flag(self.skip), None

    def end(self, token) -> None:
        if token is None: return
        mode, e0, dev = token
        e1 = torch.cuda.Event(enable_timing=True); e1.record(torch.cuda.current_stream(dev))
        self.pending.append((mode, e0, e1))


_tuners: dict = {}
def supports_per_pass() -> int:
    """Supports the fused reconstruction kernels take in one pass (more: passes carrying the running minimum; the single-node loss path: unsupported)."""
    return int(_lib.lib.smd_image_recon_supports_per_pass())


def _stale_table(ctx) -> int:
    """FLAGS['bwd_no_live'] when a launch-shape knob changed since the forward that filled the liveness table of this node's packed buffer: the backward
    re-derives the forward's strip partition from the knobs in force when it runs, and a table read with another partition calls live waves dead."""
    return FLAGS['bwd_no_live'] if getattr(ctx, 'knob_epoch', _lib.knob_epoch) != _lib.knob_epoch else 0


def row_skip_tuner(device) -> _RowSkipTuner:
    return _tuners.setdefault(torch.device(device).index, _RowSkipTuner())


class _ImageRecon(torch.autograd.Function):
    """Fused `handlers.image_recon` (src/core/handlers.py:14-67)."""

    @staticmethod
    def forward(ctx, depth, tgt, supp, T, K, K_inv, noise, seed, flags, want_warp, want_err, prepared):
        S, b, h, w = depth.shape  # always 4-D here: `image_recon_fused` squeezes the channel dim as an autograd view
        n = supp.shape[0]
        tgt_in, supp_in = tgt, supp
        depth = _check('depth', depth, (S, b, h, w)); tgt = _check('imgs', tgt, (b, 3, h, w))
        supp = _check('supp_imgs', supp, (n, b, 3, h, w)); T = _check('Ts', T, (n, b, 4, 4))
        K = _check('Ks', K, (b, 4, 4)); K_inv = _check('K_inv', K_inv, (b, 4, 4))
        if noise is not None: noise = _check('noise', noise.reshape(S, b, h, w), (S, b, h, w))
        dev = depth.device
        err = torch.empty((S, b, 1, h, w), device=dev, dtype=torch.float32) if _need_err(want_err, n) else None
        sel = torch.empty((S, b, 1, h, w), device=dev, dtype=torch.uint8)
        loss = torch.empty((), device=dev, dtype=torch.float32)
        warp0 = torch.empty((n, b, 3, h, w), device=dev, dtype=torch.float32) if want_warp else None
        nbytes = _lib.lib.smd_image_recon_workspace_bytes(b, n, S, h, w)
        ws = torch.empty(nbytes, device=dev, dtype=torch.uint8)
        # padded RGB texels of the supports + the target's SSIM window sums; written by the prep launch, reused by the backward
        supp_pk, cflags = _packed_for(prepared, tgt_in, supp_in, flags, None, None, b, n, h, w, dev)
        call('smd_image_recon_fwd', depth.data_ptr(), tgt.data_ptr(), supp.data_ptr(), T.data_ptr(), K.data_ptr(), K_inv.data_ptr(),
             noise.data_ptr() if noise is not None else None, int(seed) & (2**64 - 1), supp_pk.data_ptr(), err.data_ptr() if err is not None else None, sel.data_ptr(), loss.data_ptr(),
             warp0.data_ptr() if want_warp else None, ws.data_ptr(), nbytes, b, n, S, h, w, cflags, _stream())
        ctx.save_for_backward(depth, tgt, supp_pk, T, K, K_inv, sel)
        ctx.meta = (b, n, S, h, w, int(flags))
        ctx.knob_epoch = _lib.knob_epoch
        ctx.need_k = bool(ctx.needs_input_grad[4] or ctx.needs_input_grad[5])
        ctx.mark_non_differentiable(*([err, sel] if err is not None else [sel]))
        if want_warp: ctx.mark_non_differentiable(warp0)
        return loss, err, sel, warp0

    @staticmethod
    def backward(ctx, g_loss, *_):
        depth, tgt, supp_pk, T, K, K_inv, sel = ctx.saved_tensors
        b, n, S, h, w, flags = ctx.meta
        dev = _on(depth)
        g_loss = g_loss.to(torch.float32).contiguous()
        g_depth = torch.empty((S, b, h, w), device=dev, dtype=torch.float32)
        g_T = torch.empty((n, b, 4, 4), device=dev, dtype=torch.float32)
        g_K = torch.empty((b, 4, 4), device=dev, dtype=torch.float32) if ctx.need_k else None
        g_Ki = torch.empty((b, 4, 4), device=dev, dtype=torch.float32) if ctx.need_k else None
        if ctx.need_k: flags |= FLAGS['need_k_grad']
        nbytes = _lib.lib.smd_image_recon_workspace_bytes(b, n, S, h, w)
        ws = torch.empty(nbytes, device=dev, dtype=torch.uint8)
        tuner = row_skip_tuner(dev); tflag, token = tuner.begin(dev)
        call('smd_image_recon_bwd', depth.data_ptr(), tgt.data_ptr(), supp_pk.data_ptr(), T.data_ptr(), K.data_ptr(), K_inv.data_ptr(),
             sel.data_ptr(), g_loss.data_ptr(), g_depth.data_ptr(), g_T.data_ptr(),
             g_K.data_ptr() if ctx.need_k else None, g_Ki.data_ptr() if ctx.need_k else None,
             ws.data_ptr(), nbytes, b, n, S, h, w, flags | tflag | _stale_table(ctx), _stream())
        tuner.end(token)
        return g_depth, None, None, g_T, (g_K if ctx.needs_input_grad[4] else None), (g_Ki if ctx.needs_input_grad[5] else None), None, None, None, None, None, None


def image_recon_fused(depth, imgs, supp_imgs, Ts, Ks, K_inv=None, *, flags: int, noise=None, seed: int = 0, want_warp: bool = False,
                      want_err: bool = True, prepared: PreparedFrames | None = None):
    """depth (S,b,1,h,w)|(S,b,h,w); returns (loss, err (S,b,1,h,w)|None, sel uint8 (S,b,1,h,w), warp0 (n,b,3,h,w)|None).

    `want_err=False` (the handlers' choice: nothing on the training path reads the error map) saves its store in the kernel.

    `K_inv=None` inverts `Ks` with torch (differentiable), as `ViewSynth.forward` does (src/tools/geometry.py:383).
    `prepared`: the frame-only buffer from `image_recon_prep(imgs, supp_imgs, flags=flags)` (built without `pyramid`)."""
    if K_inv is None: K_inv = torch.linalg.inv(Ks) if Ks.requires_grad else inv_intrinsics(Ks)
    was5 = depth.ndim == 5
    d4 = depth.squeeze(2) if was5 else depth
    return _ImageRecon.apply(d4, imgs, supp_imgs, Ts, Ks, K_inv, noise, seed, flags, want_warp, want_err, prepared)


class _ImageReconDisp(torch.autograd.Function):
    """K0 fused into `handlers.image_recon`: from the network's multi-scale sigmoid disparity straight to
    the loss — `forward_postprocess`' up-sampling + `to_scaled` / `to_inv` (src/core/trainer.py:316-321) happens inside the fused
    kernel, which also writes `depth_up` for the backward and for `fwd['depth_up']`."""

    @staticmethod
    def forward(ctx, tgt, supp, T, K, K_inv, noise, seed, flags, want_warp, want_err, min_depth, max_depth, prepared, *disps):
        b, _, h, w = tgt.shape
        n, S = supp.shape[0], len(disps)
        tgt_in, supp_in = tgt, supp
        tgt = _check('imgs', tgt, (b, 3, h, w)); supp = _check('supp_imgs', supp, (n, b, 3, h, w)); T = _check('Ts', T, (n, b, 4, 4))
        K = _check('Ks', K, (b, 4, 4)); K_inv = _check('K_inv', K_inv, (b, 4, 4))
        disps = [_check(f'disp[{i}]', d) for i, d in enumerate(disps)]
        for d in disps:
            if d.ndim != 4 or d.shape[0] != b or d.shape[1] != 1: raise ValueError(f'disparities must be (b,1,hs,ws), got {tuple(d.shape)}')
        if noise is not None: noise = _check('noise', noise.reshape(S, b, h, w), (S, b, h, w))
        hs, ws = [d.shape[2] for d in disps], [d.shape[3] for d in disps]
        dev = tgt.device
        depth_up = torch.empty((S, b, 1, h, w), device=dev, dtype=torch.float32)
        err = torch.empty((S, b, 1, h, w), device=dev, dtype=torch.float32) if _need_err(want_err, n) else None
        sel = torch.empty((S, b, 1, h, w), device=dev, dtype=torch.uint8)
        loss = torch.empty((), device=dev, dtype=torch.float32)
        warp0 = torch.empty((n, b, 3, h, w), device=dev, dtype=torch.float32) if want_warp else None
        nbytes = _lib.lib.smd_image_recon_workspace_bytes(b, n, S, h, w)
        wsp = torch.empty(nbytes, device=dev, dtype=torch.uint8)
        packed, cflags = _packed_for(prepared, tgt_in, supp_in, flags, hs, ws, b, n, h, w, dev)
        call('smd_image_recon_disp_fwd', ptr_array([d.data_ptr() for d in disps]), int_array(hs), int_array(ws), S, float(min_depth or 0), float(max_depth or 0),
             tgt.data_ptr(), supp.data_ptr(), T.data_ptr(), K.data_ptr(), K_inv.data_ptr(), noise.data_ptr() if noise is not None else None,
             int(seed) & (2**64 - 1), packed.data_ptr(), depth_up.data_ptr(), err.data_ptr() if err is not None else None, sel.data_ptr(), loss.data_ptr(),
             warp0.data_ptr() if want_warp else None, wsp.data_ptr(), nbytes, b, n, h, w, cflags, _stream())
        ctx.save_for_backward(depth_up, packed, T, K, K_inv, sel)
        # `depth_up` is a differentiable output that usually has no other consumer: without this autograd would hand the backward
        # a materialised zero tensor for it (one more (S,b,h,w) read, and no dead-row skipping on the last support pass)
        ctx.set_materialize_grads(False)
        ctx.meta = (b, n, S, h, w, int(flags), hs, ws, float(min_depth or 0), float(max_depth or 0))
        ctx.need_k = bool(ctx.needs_input_grad[3] or ctx.needs_input_grad[4])
        ctx.knob_epoch = _lib.knob_epoch
        ctx.mark_non_differentiable(*([err, sel] if err is not None else [sel]))
        if want_warp: ctx.mark_non_differentiable(warp0)
        return loss, err, sel, warp0, depth_up

    @staticmethod
    def backward(ctx, g_loss, _ge, _gs, _gw, g_depth_up):
        depth_up, packed, T, K, K_inv, sel = ctx.saved_tensors
        b, n, S, h, w, flags, hs, ws, mn, mx = ctx.meta
        dev = _on(depth_up)
        g_loss = (g_loss if g_loss is not None else torch.zeros((), device=dev)).to(torch.float32).contiguous()
        if g_depth_up is not None: g_depth_up = _check('grad(depth_up)', g_depth_up.reshape(S, b, h, w), (S, b, h, w))
        g_disps = [torch.empty((b, 1, hs[s], ws[s]), device=dev, dtype=torch.float32) for s in range(S)]
        g_T = torch.empty((n, b, 4, 4), device=dev, dtype=torch.float32)
        g_K = torch.empty((b, 4, 4), device=dev, dtype=torch.float32) if ctx.need_k else None
        g_Ki = torch.empty((b, 4, 4), device=dev, dtype=torch.float32) if ctx.need_k else None
        if ctx.need_k: flags |= FLAGS['need_k_grad']
        hs_a, ws_a = int_array(hs), int_array(ws)
        nbytes = _lib.lib.smd_image_recon_disp_workspace_bytes(hs_a, ws_a, S, b, n, h, w)
        wsp = torch.empty(nbytes, device=dev, dtype=torch.uint8)
        tuner = row_skip_tuner(dev); tflag, token = tuner.begin(dev)
        call('smd_image_recon_disp_bwd', hs_a, ws_a, S, mn, mx, depth_up.data_ptr(), packed.data_ptr(), T.data_ptr(), K.data_ptr(), K_inv.data_ptr(),
             sel.data_ptr(), g_loss.data_ptr(), g_depth_up.data_ptr() if g_depth_up is not None else None,
             ptr_array([g.data_ptr() for g in g_disps]), g_T.data_ptr(), g_K.data_ptr() if ctx.need_k else None, g_Ki.data_ptr() if ctx.need_k else None,
             wsp.data_ptr(), nbytes, b, n, h, w, flags | tflag | _stale_table(ctx), _stream())
        tuner.end(token)
        return (None, None, g_T, (g_K if ctx.needs_input_grad[3] else None), (g_Ki if ctx.needs_input_grad[4] else None),
                None, None, None, None, None, None, None, None, *g_disps)


def image_recon_fused_disp(disps, imgs, supp_imgs, Ts, Ks, K_inv=None, *, flags: int, min_depth=None, max_depth=None, noise=None, seed: int = 0,
                           want_warp: bool = False, want_err: bool = True, prepared: PreparedFrames | None = None):
    """disps: sequence of (b,1,hs,ws) sigmoid disparities -> (loss, err|None, sel, warp0|None, depth_up (S,b,1,h,w)).

    The K0-fused form of `disp_to_depth` + `image_recon_fused`: one prep launch (or none with `prepared` =
    `image_recon_prep(imgs, supp_imgs, flags=flags, pyramid=[d.shape[-2:] for d in disps])`) and one fused launch that also reduces the loss."""
    if min_depth is not None and min_depth <= 0: raise ValueError(f'Min depth must be greater than 0. ({min_depth})')
    if max_depth and min_depth and max_depth < min_depth: raise ValueError(f'Max depth must be greater than min. ({max_depth} vs. {min_depth})')
    if K_inv is None: K_inv = torch.linalg.inv(Ks) if Ks.requires_grad else inv_intrinsics(Ks)
    return _ImageReconDisp.apply(imgs, supp_imgs, Ts, Ks, K_inv, noise, seed, flags, want_warp, want_err, min_depth, max_depth, prepared, *disps)


# ---------------------------------------------------------------------------------------------------
class _DispSmooth(torch.autograd.Function):
    """Fused `handlers.disp_smooth` (src/core/handlers.py:262-281) over every scale."""

    @staticmethod
    def forward(ctx, img, flags, keys, want_aux, prepared, *disps):
        b, _, h, w = img.shape
        img_in = img
        img = _check('imgs', img, (b, 3, h, w))
        disps = [_check(f'disp[{i}]', d) for i, d in enumerate(disps)]
        for d in disps:
            if d.ndim != 4 or d.shape[0] != b or d.shape[1] != 1: raise ValueError(f'disparities must be (b,1,hs,ws), got {tuple(d.shape)}')
        S = len(disps)
        hs, ws = [d.shape[2] for d in disps], [d.shape[3] for d in disps]
        dev = img.device
        loss = torch.empty((), device=dev, dtype=torch.float32)
        stats = torch.empty((S, b, 2), device=dev, dtype=torch.float32)
        aux = want_aux and keys[0] == 0
        dg = torch.empty((b, 1, hs[0], ws[0]), device=dev, dtype=torch.float32) if aux else None
        ig = torch.empty((b, 1, hs[0], ws[0]), device=dev, dtype=torch.float32) if aux else None
        hs_a, ws_a, keys_a = int_array(hs), int_array(ws), int_array(keys)
        nbytes = _lib.lib.smd_disp_smooth_workspace_bytes(hs_a, ws_a, S, b)
        wsp = torch.empty(max(nbytes, 256), device=dev, dtype=torch.uint8)
        # the edge weights depend on the frame alone: taken from `prepared` when it carries them for this frame and pyramid (filled on a side
        # stream under the networks), otherwise the forward call fills a fresh buffer first; the backward reads them instead of the image
        ew, cflags = None, int(flags)
        if cflags & FLAGS['use_edges']:
            if prepared is not None and not (cflags & FLAGS['use_laplacian']): ew = prepared.edges_for(img_in, hs, ws)
            if ew is not None:
                cur = torch.cuda.current_stream(dev)
                cur.wait_event(prepared.event); ew.record_stream(cur)
                cflags |= FLAGS['edges_ready']
            else: ew = torch.empty(_lib.lib.smd_disp_smooth_edge_weight_bytes(hs_a, ws_a, S, b), device=dev, dtype=torch.uint8)
        call('smd_disp_smooth_fwd', ptr_array([d.data_ptr() for d in disps]), hs_a, ws_a, keys_a, S, b, img.data_ptr(), h, w, cflags,
             loss.data_ptr(), stats.data_ptr(), dg.data_ptr() if aux else None, ig.data_ptr() if aux else None,
             ew.data_ptr() if ew is not None else None, wsp.data_ptr(), nbytes, _stream())
        ctx.save_for_backward(img, stats, ew, *disps)
        ctx.meta = (hs, ws, list(keys), S, b, h, w, int(flags))
        if aux: ctx.mark_non_differentiable(dg, ig)
        return loss, dg, ig

    @staticmethod
    def backward(ctx, g_loss, *_):
        img, stats, ew, *disps = ctx.saved_tensors
        _on(img)
        hs, ws, keys, S, b, h, w, flags = ctx.meta
        g_loss = g_loss.to(torch.float32).contiguous()
        g_disps = [torch.empty_like(d) for d in disps]
        call('smd_disp_smooth_bwd', ptr_array([d.data_ptr() for d in disps]), int_array(hs), int_array(ws), int_array(keys), S, b,
             img.data_ptr(), h, w, flags, stats.data_ptr(), ew.data_ptr() if ew is not None else None, g_loss.data_ptr(),
             ptr_array([g.data_ptr() for g in g_disps]), _stream())
        return (None, None, None, None, None, *g_disps)


def disp_smooth_fused(disps: dict, imgs, *, use_edges: bool = False, want_aux: bool = True, use_laplacian: bool = False, prepared: PreparedFrames | None = None):
    """disps {key: (b,1,hs,ws)} -> (loss, disp_grad|None, image_grad|None); aux maps are those of key 0.
    `use_laplacian`: second-order differences, `SmoothReg(use_laplacian=True)` (src/regularizers/smooth.py:33-48).
    `prepared`: `image_recon_prep(imgs, ..., pyramid=..., smooth_edges=True)` — its edge weights are used if they were built for `imgs` and
    this pyramid (silently ignored otherwise)."""
    keys = [int(k) for k in disps.keys()]
    flags = (FLAGS['use_edges'] if use_edges else 0) | (FLAGS['use_laplacian'] if use_laplacian else 0)
    return _DispSmooth.apply(imgs, flags, keys, want_aux, prepared, *disps.values())


# ---------------------------------------------------------------------------------------------------
class _LossPath(torch.autograd.Function):
    """`forward_loss` of the kbr configuration as ONE autograd node (round 5): `handlers.image_recon` (K0 fused) + `handlers.disp_smooth`
    (first-order, edge-aware) + the weighted sum (src/core/trainer.py:383-392, 436-437, 462-464), and in the backward the chain rule through
    the pose / intrinsics prologue (:250-262) when its leaves are given.  `smd_loss_path_fwd/_bwd`: 1 + 3 launches."""

    @staticmethod
    def forward(ctx, tgt, supp, T, K, K_inv, aa, t, invert, fs, cs, seed, flags, min_depth, max_depth, keys, prepared, w_rec, w_sm, *disps):
        b, _, h, w = tgt.shape
        n, S = supp.shape[0], len(disps)
        tgt_in, supp_in = tgt, supp
        tgt = _check('imgs', tgt, (b, 3, h, w)); supp = _check('supp_imgs', supp, (n, b, 3, h, w)); T = _check('Ts', T, (n, b, 4, 4))
        K = _check('Ks', K, (b, 4, 4)); K_inv = _check('K_inv', K_inv, (b, 4, 4))
        disps = [_check(f'disp[{i}]', d) for i, d in enumerate(disps)]
        for d in disps:
            if d.ndim != 4 or d.shape[0] != b or d.shape[1] != 1: raise ValueError(f'disparities must be (b,1,hs,ws), got {tuple(d.shape)}')
        if aa is not None:
            aa = _check('aa', aa, (n*b, 3)); t = _check('t', t, (n*b, 3))
            if invert is not None and (invert.dtype != torch.uint8 or tuple(invert.shape) != (n*b,)): raise ValueError('invert must be uint8 (n*b,)')
        if fs is not None: fs = _check('fs', fs, (b, 2)); cs = _check('cs', cs, (b, 2))
        hs, ws = [d.shape[2] for d in disps], [d.shape[3] for d in disps]
        hs_a, ws_a, keys_a = int_array(hs), int_array(ws), int_array(keys)
        dev = tgt.device
        depth_up = torch.empty((S, b, 1, h, w), device=dev, dtype=torch.float32)
        sel = torch.empty((S, b, 1, h, w), device=dev, dtype=torch.uint8)
        loss3 = torch.empty(3, device=dev, dtype=torch.float32)
        stats = torch.empty((S, b, 2), device=dev, dtype=torch.float32)
        nbytes = _lib.lib.smd_loss_path_workspace_bytes(hs_a, ws_a, S, b, n, h, w)
        wsp = torch.empty(max(nbytes, 256), device=dev, dtype=torch.uint8)
        packed, cflags = _packed_for(prepared, tgt_in, supp_in, flags, hs, ws, b, n, h, w, dev)
        cflags |= FLAGS['use_edges']
        ew = prepared.edges_for(tgt_in, hs, ws) if prepared is not None else None
        if ew is not None:
            ew.record_stream(torch.cuda.current_stream(dev))     # (the wait for `prepared.event` happened in _packed_for)
            cflags |= FLAGS['edges_ready']
        else: ew = torch.empty(_lib.lib.smd_disp_smooth_edge_weight_bytes(hs_a, ws_a, S, b), device=dev, dtype=torch.uint8)
        call('smd_loss_path_fwd', ptr_array([d.data_ptr() for d in disps]), hs_a, ws_a, keys_a, S, float(min_depth or 0), float(max_depth or 0),
             tgt.data_ptr(), supp.data_ptr(), T.data_ptr(), K.data_ptr(), K_inv.data_ptr(), int(seed) & (2**64 - 1), packed.data_ptr(), ew.data_ptr(),
             depth_up.data_ptr(), sel.data_ptr(), loss3.data_ptr(), stats.data_ptr(), wsp.data_ptr(), nbytes, b, n, h, w, cflags, float(w_rec), float(w_sm), _stream())
        ctx.save_for_backward(depth_up, packed, T, K, K_inv, sel, stats, ew, aa, t, invert, fs, cs, *disps)
        ctx.set_materialize_grads(False)
        ctx.meta = (b, n, S, h, w, int(flags), hs, ws, list(keys), float(min_depth or 0), float(max_depth or 0), float(w_rec), float(w_sm))
        ctx.need_k = bool(fs is not None or ctx.needs_input_grad[3] or ctx.needs_input_grad[4])
        ctx.knob_epoch = _lib.knob_epoch
        total, l_rec, l_sm = loss3[0], loss3[1], loss3[2]
        ctx.mark_non_differentiable(l_rec, l_sm, sel)
        return total, l_rec, l_sm, sel, depth_up

    @staticmethod
    def backward(ctx, g_loss, _g1, _g2, _gs, g_depth_up):
        depth_up, packed, T, K, K_inv, sel, stats, ew, aa, t, invert, fs, cs, *disps = ctx.saved_tensors
        b, n, S, h, w, flags, hs, ws, keys, mn, mx, w_rec, w_sm = ctx.meta
        dev = _on(depth_up)
        if g_depth_up is not None: raise NotImplementedError('loss_path_fused: `depth_up` has another differentiable consumer; use image_recon_fused_disp + disp_smooth_fused')
        g_loss = (g_loss if g_loss is not None else torch.zeros((), device=dev)).to(torch.float32).contiguous()
        g_disps = [torch.empty((b, 1, hs[s], ws[s]), device=dev, dtype=torch.float32) for s in range(S)]
        g_T = torch.empty((n, b, 4, 4), device=dev, dtype=torch.float32)
        g_K = torch.empty((b, 4, 4), device=dev, dtype=torch.float32) if ctx.need_k else None
        g_Ki = torch.empty((b, 4, 4), device=dev, dtype=torch.float32) if ctx.need_k else None
        g_aa = torch.empty_like(aa) if aa is not None else None
        g_t = torch.empty_like(t) if aa is not None else None
        g_fs = torch.empty_like(fs) if fs is not None else None
        g_cs = torch.empty_like(cs) if fs is not None else None
        cflags = flags | FLAGS['use_edges'] | (FLAGS['need_k_grad'] if ctx.need_k else 0)
        hs_a, ws_a, keys_a = int_array(hs), int_array(ws), int_array(keys)
        nbytes = _lib.lib.smd_loss_path_workspace_bytes(hs_a, ws_a, S, b, n, h, w)
        wsp = torch.empty(max(nbytes, 256), device=dev, dtype=torch.uint8)
        tuner = row_skip_tuner(dev); tflag, token = tuner.begin(dev)
        P = lambda x: x.data_ptr() if x is not None else None
        call('smd_loss_path_bwd', ptr_array([d.data_ptr() for d in disps]), hs_a, ws_a, keys_a, S, mn, mx, depth_up.data_ptr(), packed.data_ptr(), T.data_ptr(),
             K.data_ptr(), K_inv.data_ptr(), sel.data_ptr(), stats.data_ptr(), ew.data_ptr(), g_loss.data_ptr(), w_rec, w_sm,
             P(aa), P(t), P(invert), P(fs), P(cs), ptr_array([g.data_ptr() for g in g_disps]), g_T.data_ptr(), P(g_K), P(g_Ki), P(g_aa), P(g_t), P(g_fs), P(g_cs),
             wsp.data_ptr(), nbytes, b, n, h, w, cflags | tflag | _stale_table(ctx), _stream())
        tuner.end(token)
        need = ctx.needs_input_grad
        return (None, None, (g_T if need[2] else None), (g_K if need[3] else None), (g_Ki if need[4] else None), g_aa, g_t, None, g_fs, g_cs,
                None, None, None, None, None, None, None, None, *g_disps)


def loss_path_fused(disps: dict, imgs, supp_imgs, Ts, Ks, K_inv=None, *, pose=None, intrinsics=None, flags: int, min_depth=None, max_depth=None,
                    seed: int = 0, w_recon: float = 1.0, w_smooth: float = 0.001, prepared: PreparedFrames | None = None):
    """`forward_loss` with `img_recon` + `disp_smooth(use_edges=True)` as one operator:
        -> (loss = w_recon*l_recon + w_smooth*l_smooth, l_recon, l_smooth, sel (S,b,1,h,w) uint8, depth_up (S,b,1,h,w)).

    disps {key: (b,1,hs,ws)} sigmoid disparities (key = the `s` of `loss_s / 2**s`); Ts (n,b,4,4), Ks (b,4,4) [, K_inv].
    `pose=(aa, t, invert)`: the (n*b,3) leaves `Ts` was built from with `pose_matrices` — then `Ts` is taken as a value and the backward hands the
    gradients to `aa` and `t` directly (no `pose_matrices` backward launch); likewise `intrinsics=(fs, cs)` for `Ks`, `K_inv` from `intrinsics`.
    Raises `_lib.Unsupported` for what the operator does not serve (see include/smd_hotpath.h); `depth_up` must not have another
    differentiable consumer."""
    if min_depth is not None and min_depth <= 0: raise ValueError(f'Min depth must be greater than 0. ({min_depth})')
    if max_depth and min_depth and max_depth < min_depth: raise ValueError(f'Max depth must be greater than min. ({max_depth} vs. {min_depth})')
    aa, t, inv = pose if pose is not None else (None, None, None)
    fs, cs = intrinsics if intrinsics is not None else (None, None)
    if pose is not None: Ts = Ts.detach()
    if intrinsics is not None:
        if K_inv is None: raise ValueError('intrinsics=(fs, cs) goes with the K, K_inv that `functional.intrinsics(fs, cs, size)` returned')
        if pose is None:    # the intrinsics' chain rule rides on the pose chain's guest block (smd_loss_path_bwd): without it the backward would fail, after a forward that succeeded
            raise _lib.Unsupported('intrinsics=(fs, cs) needs pose=(aa, t, invert): pass K, K_inv alone and let autograd carry their gradients')
        Ks, K_inv = Ks.detach(), K_inv.detach()
    if K_inv is None: K_inv = torch.linalg.inv(Ks) if Ks.requires_grad else inv_intrinsics(Ks)
    keys = [int(k) for k in disps.keys()]
    return _LossPath.apply(imgs, supp_imgs, Ts, Ks, K_inv, aa, t, inv, fs, cs, seed, flags, min_depth, max_depth, keys, prepared, w_recon, w_smooth, *disps.values())


# ---------------------------------------------------------------------------------------------------
# Un-fused, class-level operators
# ---------------------------------------------------------------------------------------------------
class _ViewSynth(torch.autograd.Function):
    """`ViewSynth.forward` (src/tools/geometry.py:366-391) for any channel count."""

    @staticmethod
    def forward(ctx, inp, depth, T, K, K_inv):
        B, Cc, h, w = inp.shape
        inp = _check('input', inp, (B, Cc, h, w)); depth = _check('depth', depth, (B, 1, h, w))
        T = _check('T', T, (B, 4, 4)); K = _check('K', K, (B, 4, 4)); K_inv = _check('K_inv', K_inv, (B, 4, 4))
        warp = torch.empty_like(inp)
        dwarp = torch.empty((B, 1, h, w), device=inp.device, dtype=torch.float32)
        valid = torch.empty((B, 1, h, w), device=inp.device, dtype=torch.uint8)
        call('smd_view_synth_fwd', inp.data_ptr(), depth.data_ptr(), T.data_ptr(), K.data_ptr(), K_inv.data_ptr(), warp.data_ptr(),
             dwarp.data_ptr(), valid.data_ptr(), B, Cc, h, w, _stream())
        ctx.save_for_backward(inp, depth, T, K, K_inv)
        ctx.mark_non_differentiable(valid)
        return warp, dwarp, valid

    @staticmethod
    def backward(ctx, g_warp, g_dwarp, _g_valid):
        inp, depth, T, K, K_inv = ctx.saved_tensors
        B, Cc, h, w = inp.shape
        dev = _on(inp)
        g_warp = _check('grad(warp)', g_warp if g_warp is not None else torch.zeros_like(inp))
        g_dwarp = _check('grad(depth_warp)', g_dwarp) if g_dwarp is not None else None
        need_in, need_k = ctx.needs_input_grad[0], (ctx.needs_input_grad[3] or ctx.needs_input_grad[4])
        g_in = torch.empty_like(inp) if need_in else None
        g_depth = torch.empty_like(depth)
        g_T = torch.empty((B, 4, 4), device=dev, dtype=torch.float32)
        g_K = torch.empty((B, 4, 4), device=dev, dtype=torch.float32) if need_k else None
        g_Ki = torch.empty((B, 4, 4), device=dev, dtype=torch.float32) if need_k else None
        nbytes = _lib.lib.smd_view_synth_workspace_bytes(B, h, w)
        ws = torch.empty(nbytes, device=dev, dtype=torch.uint8)
        call('smd_view_synth_bwd', inp.data_ptr(), depth.data_ptr(), T.data_ptr(), K.data_ptr(), K_inv.data_ptr(), g_warp.data_ptr(),
             g_dwarp.data_ptr() if g_dwarp is not None else None, g_in.data_ptr() if need_in else None, g_depth.data_ptr(), g_T.data_ptr(),
             g_K.data_ptr() if need_k else None, g_Ki.data_ptr() if need_k else None, ws.data_ptr(), nbytes, B, Cc, h, w, _stream())
        return g_in, g_depth, g_T, (g_K if ctx.needs_input_grad[3] else None), (g_Ki if ctx.needs_input_grad[4] else None)


def view_synth(inp, depth, T, K, K_inv=None):
    """-> (input_warp (B,C,h,w), depth_warp (B,1,h,w), mask_valid (B,1,h,w) bool)."""
    if K_inv is None: K_inv = torch.linalg.inv(K)
    warp, dwarp, valid = _ViewSynth.apply(inp, depth, T, K, K_inv)
    return warp, dwarp, valid.bool()


class _PhotoError(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, flags, weight_ssim):
        if pred.ndim != 4: raise ValueError(f'photometric error expects (N,C,h,w) tensors, got {tuple(pred.shape)}')
        N, c, h, w = pred.shape
        pred = _check('pred', pred, (N, c, h, w)); target = _check('target', target, (N, c, h, w))
        err = torch.empty((N, 1, h, w), device=pred.device, dtype=torch.float32)
        call('smd_photo_error_fwd', pred.data_ptr(), target.data_ptr(), err.data_ptr(), N, c, h, w, int(flags), float(weight_ssim), _stream())
        ctx.save_for_backward(pred, target); ctx.flags, ctx.weight_ssim = int(flags), float(weight_ssim)
        return err

    @staticmethod
    def backward(ctx, g_err):
        pred, target = ctx.saved_tensors
        _on(pred)
        N, c, h, w = pred.shape
        g_err = _check('grad(err)', g_err)
        g_pred = torch.empty_like(pred)
        nbytes = _lib.lib.smd_photo_error_workspace_bytes(N, c, h, w)
        ws = torch.empty(nbytes, device=pred.device, dtype=torch.uint8)
        call('smd_photo_error_bwd', pred.data_ptr(), target.data_ptr(), g_err.data_ptr(), g_pred.data_ptr(), ws.data_ptr(), nbytes, N, c, h, w,
             ctx.flags, ctx.weight_ssim, _stream())
        return g_pred, None, None, None


def photo_error(pred, target, loss_name: str = 'ssim', weight_ssim: float = 0.85):
    """(N,C,h,w) x2 -> (N,1,h,w): weight_ssim * SSIM + (1 - weight_ssim) * L1 ('ssim'; `PhotoError(weight_ssim)`,
    src/losses/photometric.py:65-88), channel-mean |.| ('l1') or Euclidean distance ('l2')."""
    if loss_name not in ('ssim', 'l1', 'l2'): raise KeyError(loss_name)
    if not (0 <= weight_ssim <= 1): raise ValueError(f'Invalid SSIM weight. ({weight_ssim} vs. [0, 1])')
    return _PhotoError.apply(pred, target, {'ssim': 0, 'l1': FLAGS['loss_l1'], 'l2': FLAGS['loss_l2']}[loss_name], weight_ssim)


class _Regression(torch.autograd.Function):
    """`RegressionLoss.forward` (src/losses/regression.py:69-75); gradients to both `pred` and `target`."""

    @staticmethod
    def forward(ctx, pred, target, mask, flags):
        pred = _check('pred', pred); target = _check('target', target, pred.shape)
        if mask is not None:
            if tuple(mask.shape) != tuple(pred.shape): raise ValueError(f'mask: expected shape {tuple(pred.shape)}, got {tuple(mask.shape)}')
            # The reference multiplies by the mask (`mask*err`, `err.sum()/mask.sum()`, src/losses/regression.py:72-74), so a float mask
            # there is a per-pixel WEIGHT; the kernel implements the 0/1 case every caller on this path uses (automask, validity).
            if mask.dtype.is_floating_point: raise TypeError('RegressionLoss: pass a bool (or uint8 0/1) mask; weighting masks are not part of the accelerated path')
            mask = (mask if mask.dtype == torch.bool else mask != 0).contiguous().view(torch.uint8)
        N, dev = pred.numel(), pred.device
        loss = torch.empty((), device=dev, dtype=torch.float32); err = torch.empty_like(pred)
        stats = torch.zeros(8, device=dev, dtype=torch.float32)
        nbytes = _lib.lib.smd_regression_workspace_bytes(N)
        ws = torch.empty(nbytes, device=dev, dtype=torch.uint8)
        call('smd_regression_fwd', pred.data_ptr(), target.data_ptr(), mask.data_ptr() if mask is not None else None, N, int(flags),
             loss.data_ptr(), err.data_ptr(), stats.data_ptr(), ws.data_ptr(), nbytes, _stream())
        ctx.save_for_backward(pred, target, mask, stats); ctx.flags = int(flags)
        ctx.mark_non_differentiable(err)
        return loss, err

    @staticmethod
    def backward(ctx, g_loss, _g_err):
        pred, target, mask, stats = ctx.saved_tensors
        _on(pred)
        N = pred.numel()
        g_pred = torch.empty_like(pred) if ctx.needs_input_grad[0] else None
        g_target = torch.empty_like(target) if ctx.needs_input_grad[1] else None
        if g_pred is None and g_target is None: return None, None, None, None
        nbytes = _lib.lib.smd_regression_workspace_bytes(N)
        ws = torch.empty(nbytes, device=pred.device, dtype=torch.uint8)
        call('smd_regression_bwd', pred.data_ptr(), target.data_ptr(), mask.data_ptr() if mask is not None else None, N, ctx.flags,
             stats.data_ptr(), g_loss.to(torch.float32).contiguous().data_ptr(), g_pred.data_ptr() if g_pred is not None else None,
             g_target.data_ptr() if g_target is not None else None, ws.data_ptr(), nbytes, _stream())
        return g_pred, g_target, None, None


def regression_loss(pred, target, mask=None, *, loss_name: str = 'berhu', invert: bool = False):
    """Masked mean of a dense regression error -> (loss, err).  loss_name in {'l1', 'log_l1', 'berhu'}."""
    if loss_name not in ('l1', 'log_l1', 'berhu'): raise KeyError(loss_name)
    return _Regression.apply(pred, target, mask, REGR_FLAGS[loss_name] | (REGR_FLAGS['invert'] if invert else 0))


class _ReconReduce(torch.autograd.Function):
    @staticmethod
    def forward(ctx, err_warp, err_static, mask, noise, seed, flags):
        n, B, h, w = err_warp.shape
        err_warp = _check('err_warp', err_warp, (n, B, h, w))
        if err_static is not None: err_static = _check('err_static', err_static, (n, B, h, w))
        if mask is not None: mask = _check('mask', mask, (B, n, h, w))
        if noise is not None: noise = _check('noise', noise.reshape(B, h, w), (B, h, w))
        dev = err_warp.device
        err = torch.empty((B, h, w), device=dev, dtype=torch.float32)
        sel = torch.empty((B, h, w), device=dev, dtype=torch.uint8)
        loss = torch.empty((), device=dev, dtype=torch.float32)
        nbytes = _lib.lib.smd_recon_reduce_workspace_bytes(B, h, w)
        ws = torch.empty(nbytes, device=dev, dtype=torch.uint8)
        call('smd_recon_reduce_fwd', err_warp.data_ptr(), err_static.data_ptr() if err_static is not None else None,
             mask.data_ptr() if mask is not None else None, noise.data_ptr() if noise is not None else None, int(seed) & (2**64 - 1),
             err.data_ptr(), sel.data_ptr(), loss.data_ptr(), ws.data_ptr(), nbytes, n, B, h, w, int(flags), _stream())
        if mask is not None: ctx.save_for_backward(sel, err_warp, err_static, mask)   # the masked forms' derivatives need the errors and the mask
        else: ctx.save_for_backward(sel, None, None, None)
        ctx.meta = (n, B, h, w, int(flags))
        ctx.mark_non_differentiable(err, sel)
        return loss, err, sel

    @staticmethod
    def backward(ctx, g_loss, *_):
        sel, err_warp, err_static, mask = ctx.saved_tensors
        _on(sel)
        n, B, h, w, flags = ctx.meta
        g = torch.empty((n, B, h, w), device=sel.device, dtype=torch.float32)
        g_mask = torch.empty_like(mask) if mask is not None else None
        call('smd_recon_reduce_bwd', sel.data_ptr(), g_loss.to(torch.float32).contiguous().data_ptr(), g.data_ptr(),
             err_warp.data_ptr() if err_warp is not None else None, err_static.data_ptr() if err_static is not None else None,
             mask.data_ptr() if mask is not None else None, g_mask.data_ptr() if g_mask is not None else None, n, B, h, w, flags, _stream())
        return g, None, g_mask, None, None, None


def recon_reduce(err_warp, err_static=None, *, use_min: bool = False, noise=None, seed: int = 0, mask=None, mask_name: str | None = None):
    """Per-support error maps (n,B,h,w) [+ static ones] -> (loss, err (B,h,w), sel uint8 (B,h,w); 255 = auto-masked).

    `mask` (B,n,h,w) with `mask_name` 'explainability' | 'uncertainty': the predictive weighting of `ReconstructionLoss.apply_mask`
    (src/losses/reconstruction.py:46-57), applied to the warped and the static errors before the reductions; differentiable."""
    if mask_name not in {'explainability', 'uncertainty', None}: raise ValueError(f'Invalid mask type: {mask_name}')
    if mask_name and mask is None: raise ValueError("Must provide a 'mask' when masking...")
    flags = (FLAGS['use_min'] if use_min else 0) | (FLAGS['use_automask'] if err_static is not None else 0)
    if mask_name:
        flags |= FLAGS['mask_' + mask_name]
        if mask.shape[1] == 1 and err_warp.shape[0] > 1: mask = mask.expand(-1, err_warp.shape[0], -1, -1)   # one mask for every support (broadcast in the reference)
    return _ReconReduce.apply(err_warp, err_static, mask if mask_name else None, noise, seed, flags)


# ---------------------------------------------------------------------------------------------------
# ---------------------------------------------------------------------------------------------------
def _glue_ws(B, C, h, w, device):
    nbytes = _lib.lib.smd_decoder_glue_workspace_bytes(B, C, h, w)
    return torch.empty(nbytes, device=device, dtype=torch.uint8), nbytes


def _check_fb(name: str, t: torch.Tensor, shape=None) -> torch.Tensor:
    """Like `_check`, for the operators that also take bfloat16 tensors at an autocast boundary."""
    if not isinstance(t, torch.Tensor): raise TypeError(f'{name} must be a Tensor, got {type(t)}')
    if not t.is_cuda: raise RuntimeError(f'{name} must live on the GPU')
    if t.dtype not in (torch.float32, torch.bfloat16): raise TypeError(f'{name} must be float32 or bfloat16, got {t.dtype}')
    if shape is not None and tuple(t.shape) != tuple(shape): raise ValueError(f'{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}')
    return t.contiguous()


_BF = torch.bfloat16


class _EluPad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bias, apply_elu, out_dtype):
        x = _check_fb('x', x)
        if x.ndim != 4: raise ValueError(f'expected (B,C,h,w), got {tuple(x.shape)}')
        B, C, h, w = x.shape
        if bias is not None: bias = _check('bias', bias, (C,))
        out = torch.empty((B, C, h + 2, w + 2), device=x.device, dtype=out_dtype)
        dt = (1 if x.dtype == _BF else 0) | (4 if out_dtype == _BF else 0)
        call('smd_elu_pad_fwd', x.data_ptr(), bias.data_ptr() if bias is not None else None, out.data_ptr(), B, C, h, w, int(apply_elu), dt, _stream())
        ctx.save_for_backward(x, bias); ctx.apply_elu, ctx.dt, ctx.out_dtype = int(apply_elu), dt, out_dtype
        return out

    @staticmethod
    def backward(ctx, g_out):
        x, bias = ctx.saved_tensors
        _on(x)
        B, C, h, w = x.shape
        g_x = torch.empty_like(x)
        g_b = torch.empty_like(bias) if (bias is not None and ctx.needs_input_grad[1]) else None
        ws, nbytes = _glue_ws(B, C, h, w, x.device) if g_b is not None else (None, 0)
        call('smd_elu_pad_bwd', x.data_ptr(), bias.data_ptr() if bias is not None else None, g_out.to(ctx.out_dtype).contiguous().data_ptr(),
             g_x.data_ptr(), g_b.data_ptr() if g_b is not None else None, ws.data_ptr() if ws is not None else None, nbytes, B, C, h, w,
             ctx.apply_elu, ctx.dt, _stream())
        return g_x, g_b, None, None


def elu_pad(x, bias=None, apply_elu: bool = True, out_dtype=None):
    """reflect_pad1(elu(x + bias)) (or just bias + padding): the input of the next 3x3 convolution of the decoder.
    x float32 or bfloat16; `out_dtype` (default: x's) may be bfloat16 for a bf16 consumer; bias and arithmetic are fp32."""
    return _EluPad.apply(x, bias, apply_elu, out_dtype or x.dtype)


class _Conv3x3Head(torch.autograd.Function):
    """`act(conv3x3(xp, weight (1,C,3,3)) + bias)` on an already reflection-padded input (`smd_conv3x3_head_*`): the decoder's output heads.  xp may be bfloat16
    (the decoder under bf16 autocast): the output, the weights' gradient and every sum stay fp32, `g_xp` comes back in xp's type."""
    @staticmethod
    def forward(ctx, xp, weight, bias, act):
        xp = _check_fb('xp', xp)
        if xp.ndim != 4 or xp.shape[2] < 4 or xp.shape[3] < 4: raise ValueError(f'expected a padded (B,C,h+2,w+2) with h, w >= 2, got {tuple(xp.shape)}')
        B, C, H, W = xp.shape
        weight = _check('weight', weight, (1, C, 3, 3))
        if bias is not None: bias = _check('bias', bias, (1,))
        y = torch.empty((B, 1, H - 2, W - 2), device=xp.device, dtype=torch.float32)
        act = int(act) | (2 if xp.dtype == _BF else 0)            # SMD_HEAD_X_BF16
        call('smd_conv3x3_head_fwd', xp.data_ptr(), weight.data_ptr(), bias.data_ptr() if bias is not None else None, y.data_ptr(), B, C, H - 2, W - 2, act, _stream())
        ctx.save_for_backward(xp, weight, y); ctx.act, ctx.has_bias = act, bias is not None
        return y

    @staticmethod
    def backward(ctx, g_y):
        xp, weight, y = ctx.saved_tensors
        dev = _on(xp)
        B, C, H, W = xp.shape
        need_x, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        g_y = _check('grad(y)', g_y.float(), (B, 1, H - 2, W - 2))
        g_xp = torch.empty_like(xp) if need_x else None
        g_w = torch.empty_like(weight) if (need_w or need_b) else None
        g_b = torch.empty(1, device=dev, dtype=torch.float32) if need_b else None
        nbytes = _lib.lib.smd_conv3x3_head_workspace_bytes(B, C, H - 2, W - 2) if g_w is not None else 0
        ws = torch.empty(max(nbytes, 256), device=dev, dtype=torch.uint8) if g_w is not None else None
        if g_xp is not None or g_w is not None:
            call('smd_conv3x3_head_bwd', xp.data_ptr(), weight.data_ptr(), y.data_ptr(), g_y.data_ptr(), g_xp.data_ptr() if g_xp is not None else None,
                 g_w.data_ptr() if g_w is not None else None, g_b.data_ptr() if g_b is not None else None, ws.data_ptr() if ws is not None else None, nbytes,
                 B, C, H - 2, W - 2, ctx.act, _stream())
        return g_xp, (g_w if need_w else None), g_b, None


def conv3x3_head(xp, weight, bias=None, act: str | None = 'sigmoid'):
    """`act(F.conv2d(xp, weight, bias))` for ONE output channel and an input that is already reflection-padded (`elu_pad`'s output): the decoder's
    output heads (src/networks/decoders/monodepth.py:52, 86-87).  xp (B,C,h+2,w+2) fp32 or bf16, weight (1,C,3,3), bias (1) or None -> (B,1,h,w) fp32; act 'sigmoid' | None."""
    if act not in ('sigmoid', 'none', None): raise ValueError(f"act must be 'sigmoid' or None, got {act!r}")
    return _Conv3x3Head.apply(xp, weight, bias, 1 if act == 'sigmoid' else 0)


class _Conv3x3Thin(torch.autograd.Function):
    """`F.conv2d(xp, weight (16,C,3,3))` on an already reflection-padded input on the f32 MFMA (`smd_conv3x3_thin_*`: `v_mfma_f32_16x16x4_f32`, operands staged
    through LDS): the decoder's last stage in its round-5 form (round 6: `conv3x3_wide` routes between this and the split-bf16 form per operator)."""
    @staticmethod
    def forward(ctx, xp, weight):
        xp = _check('xp', xp)
        if xp.ndim != 4 or xp.shape[2] < 4 or xp.shape[3] < 4: raise ValueError(f'expected a padded (B,C,h+2,w+2) with h, w >= 2, got {tuple(xp.shape)}')
        B, C, H, W = xp.shape
        weight = _check('weight', weight, (16, C, 3, 3))
        y = torch.empty((B, 16, H - 2, W - 2), device=xp.device, dtype=torch.float32)
        call('smd_conv3x3_thin_fwd', xp.data_ptr(), weight.data_ptr(), y.data_ptr(), B, C, H - 2, W - 2, _stream())
        ctx.save_for_backward(xp, weight)
        return y

    @staticmethod
    def backward(ctx, g_y):
        xp, weight = ctx.saved_tensors
        dev = _on(xp)
        B, C, H, W = xp.shape
        need_x, need_w = ctx.needs_input_grad
        g_y = _check('grad(y)', g_y, (B, 16, H - 2, W - 2))
        g_xp = g_w = None
        if need_x: g_xp = torch.empty_like(xp)
        if need_w: g_w = torch.empty_like(weight)
        if need_x or need_w:
            nbytes = _lib.lib.smd_conv3x3_thin_workspace_bytes(B, C, H - 2, W - 2) if need_w else 0
            ws = torch.empty(max(nbytes, 256), device=dev, dtype=torch.uint8) if need_w else None
            call('smd_conv3x3_thin_bwd', xp.data_ptr(), weight.data_ptr(), g_y.data_ptr(), g_xp.data_ptr() if need_x else None, g_w.data_ptr() if need_w else None,
                 ws.data_ptr() if ws is not None else None, nbytes, B, C, H - 2, W - 2, _stream())
        return g_xp, g_w


def conv3x3_thin(xp, weight):
    """`F.conv2d(xp, weight)` for sixteen output channels and an input that is already reflection-padded: the thin up-convolution of the decoder's last
    stage (src/networks/decoders/monodepth.py:45-50, 80-84), bias-free (the next glue kernel adds it).  xp (B,C,h+2,w+2), weight (16,C,3,3) -> (B,16,h,w);
    C = 16 or 32 (`_lib.Unsupported` otherwise)."""
    return _Conv3x3Thin.apply(xp, weight)


# ---- the wide decoder convolutions: split-bf16 MFMA kernels (smd_conv3x3_mfma_*) or MIOpen, per operator and shape --------------------------------
# Which of the two serves an (operator, shape) pair is decided by a same-box A/B the first time the pair is seen: both run on the call's own tensors,
# interleaved, a few times each; the faster one is cached for the process ("only where a same-box A/B against MIOpen wins").
# `set_conv_route('mfma' | 'miopen')` pins the choice (tests, profiles); inside a HIP-graph capture nothing is timed and a static rule stands in.
_CONV_ROUTE_MODE = 'auto'
_CONV_ROUTES: dict = {}


def set_conv_route(mode: str = 'auto'):
    """'auto' (A/B on first use), 'mfma' or 'miopen' for every wide decoder convolution; clears the cached decisions."""
    global _CONV_ROUTE_MODE
    if mode not in ('auto', 'mfma', 'miopen'): raise ValueError(mode)
    _CONV_ROUTE_MODE = mode
    _CONV_ROUTES.clear()


def conv_routes() -> dict:
    """The decisions taken so far: {(op, B, C, CO, h, w): (use_mfma, us_mfma, us_miopen)}."""
    return dict(_CONV_ROUTES)


def _conv_static_rule(op, B, C, CO, h, w):
    """Stand-in where nothing may be timed (graph capture): the shapes that won on an MI355X at cfg 2 (profiles/r06_decoder_convs.txt)."""
    px = B*h*w
    if op.endswith('_bf16'): return CO == 16           # (bf16 tensors: MIOpen's bf16 kernels serve the wide layers; the thin stage is the stencil-like case)
    if CO == 16: return op != 'wgt'
    if op == 'fwd': return px >= 20000 and C*CO <= 128*64
    if op == 'data': return px >= 5000 and C <= 256
    return px >= 20000 and CO <= 64


def _conv_route(op, B, C, CO, h, w, run_mfma, run_ref):
    if _CONV_ROUTE_MODE != 'auto': return _CONV_ROUTE_MODE == 'mfma'
    key = (op, B, C, CO, h, w)
    r = _CONV_ROUTES.get(key)
    if r is None:
        if torch.cuda.is_current_stream_capturing(): return _conv_static_rule(op, B, C, CO, h, w)
        for _ in range(2): run_mfma(); run_ref()
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(5)]
        for e in ev:                     # interleaved: a box's clocks drift over the first milliseconds, whatever is timed first looks slower
            e[0].record(); run_mfma(); e[1].record(); run_ref(); e[2].record()
        torch.cuda.synchronize()
        t_m = sorted(e[0].elapsed_time(e[1]) for e in ev)[2]*1e3
        t_r = sorted(e[1].elapsed_time(e[2]) for e in ev)[2]*1e3
        r = _CONV_ROUTES[key] = (t_m < 0.97*t_r, t_m, t_r)
    return r[0]


def _mfma_pack(weight, C, CO, pieces, want_fwd, want_bwd):
    nbytes = _lib.lib.smd_conv3x3_mfma_packed_bytes(C, CO, pieces)
    wf = torch.empty(max(nbytes, 256), device=weight.device, dtype=torch.uint8) if want_fwd else None
    wb = torch.empty(max(nbytes, 256), device=weight.device, dtype=torch.uint8) if want_bwd else None
    call('smd_conv3x3_mfma_pack', weight.data_ptr(), wf.data_ptr() if wf is not None else None, wb.data_ptr() if wb is not None else None, C, CO, pieces, _stream())
    return wf, wb


def _mfma_ws(B, C, CO, h, w, dev):
    nws = _lib.lib.smd_conv3x3_mfma_workspace_bytes(B, C, CO, h, w)
    return torch.empty(max(nws, 256), device=dev, dtype=torch.uint8), nws


class _Conv3x3Wide(torch.autograd.Function):
    """`F.conv2d(xp, weight (CO,C,3,3))` on an already reflection-padded input; each of the three operators (forward, data gradient, weight gradient) runs
    on the bf16 matrix cores (`smd_conv3x3_mfma_*`) or through the alternative — MIOpen, or for the 16-channel last stage in fp32 the f32-MFMA kernels
    `smd_conv3x3_thin_*` — as `_conv_route` says (`force`: always the MFMA kernels).  fp32 tensors: every operand split into three bf16 pieces, fp32-class
    results.  bfloat16 tensors (the decoder under bf16 autocast): one piece, bf16 in and out, the weights as their bf16 rounding (what autocast hands a bf16
    convolution), fp32 accumulation and an fp32 weight gradient."""
    @staticmethod
    def forward(ctx, xp, weight, pieces, force):
        xp = _check_fb('xp', xp)
        if xp.ndim != 4 or xp.shape[2] < 3 or xp.shape[3] < 3: raise ValueError(f'expected a padded (B,C,h+2,w+2), got {tuple(xp.shape)}')
        B, C, H, W = xp.shape
        if weight.ndim != 4 or tuple(weight.shape[1:]) != (C, 3, 3): raise ValueError(f'weight: expected (CO,{C},3,3), got {tuple(weight.shape)}')
        CO = weight.shape[0]
        weight = _check('weight', weight, (CO, C, 3, 3))
        h, w, dev = H - 2, W - 2, xp.device
        bf = xp.dtype == _BF
        if bf: pieces = 1
        thin = CO == 16 and C in (16, 32)                   # the last stage
        fwd_ok = (C % 16 == 0 and CO % 32 == 0) or thin
        if force and not fwd_ok:
            raise _lib.Unsupported(f'the MFMA forward serves C % 16 == 0 with CO % 32 == 0, or CO = 16 with C = 16 | 32, not C={C} CO={CO}')
        bwd_form = (CO % 16 == 0 and C % 32 == 0) or (C == 16 and CO == 16)   # the data gradient's own operand order (what the backward kernel serves)
        y = torch.empty((B, CO, h, w), device=dev, dtype=xp.dtype)
        packed = {}

        def run_mfma():
            if 'wf' not in packed: packed['wf'], packed['wb'] = _mfma_pack(weight, C, CO, pieces, True, bwd_form)
            ws, nws = _mfma_ws(B, C, CO, h, w, dev)
            call('smd_conv3x3_mfma_fwd', xp.data_ptr(), packed['wf'].data_ptr(), y.data_ptr(), ws.data_ptr(), nws, B, C, CO, h, w, pieces, _stream())

        def run_ref():
            if bf: return torch.conv2d(xp, weight.to(_BF))
            if thin: call('smd_conv3x3_thin_fwd', xp.data_ptr(), weight.data_ptr(), y.data_ptr(), B, C, h, w, _stream()); return y
            return torch.conv2d(xp, weight)
        use = fwd_ok and (force or _conv_route('fwd_bf16' if bf else 'fwd', B, C, CO, h, w, run_mfma, run_ref))
        if use: run_mfma()
        else: y = run_ref()
        ctx.save_for_backward(xp, weight, packed.get('wb'))
        ctx.pieces, ctx.force = pieces, force
        return y

    @staticmethod
    def backward(ctx, g_y):
        xp, weight, wp_bwd = ctx.saved_tensors
        dev = _on(xp)
        B, C, H, W = xp.shape
        CO, pieces, force, h, w = weight.shape[0], ctx.pieces, ctx.force, H - 2, W - 2
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        bf = xp.dtype == _BF
        g_y = _check_fb('grad(y)', g_y.to(xp.dtype), (B, CO, h, w))
        g_xp = g_w = None
        thin = CO == 16 and C in (16, 32)
        w_ref = weight.to(_BF) if bf else weight
        cb = lambda mask: torch.ops.aten.convolution_backward(g_y, xp, w_ref, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, mask)
        sfx = '_bf16' if bf else ''

        def thin_bwd(want_x, want_w):                       # the f32-MFMA kernels of the last stage (smd_conv3x3_thin_bwd; fp32 tensors only)
            gx_ = torch.empty_like(xp) if want_x else None
            gw_ = torch.empty_like(weight) if want_w else None
            nb = _lib.lib.smd_conv3x3_thin_workspace_bytes(B, C, h, w) if want_w else 0
            ws_ = torch.empty(max(nb, 256), device=dev, dtype=torch.uint8) if want_w else None
            call('smd_conv3x3_thin_bwd', xp.data_ptr(), weight.data_ptr(), g_y.data_ptr(), gx_.data_ptr() if want_x else None, gw_.data_ptr() if want_w else None,
                 ws_.data_ptr() if want_w else None, nb, B, C, h, w, _stream())
            return gx_, gw_
        if need_x:
            g_xp = torch.empty_like(xp)
            packed = {'wb': wp_bwd}

            def run_data():
                if packed['wb'] is None: packed['wb'] = _mfma_pack(weight, C, CO, pieces, False, True)[1]
                ws, nws = _mfma_ws(B, C, CO, h, w, dev)
                call('smd_conv3x3_mfma_bwd_data', g_y.data_ptr(), packed['wb'].data_ptr(), g_xp.data_ptr(), ws.data_ptr(), nws, B, C, CO, h, w, pieces, _stream())
            ok = (CO % 16 == 0 and C % 32 == 0) or (C == 16 and CO == 16)
            ref_data = (lambda: thin_bwd(True, False)[0]) if (thin and not bf) else (lambda: cb([True, False, False])[0])
            if ok and (force or _conv_route('data' + sfx, B, C, CO, h, w, run_data, ref_data)): run_data()
            else: g_xp = ref_data()                         # (also: channel counts the data-gradient kernel does not tile)
        if need_w:
            g_w = torch.empty_like(weight)

            def run_wgt():
                ws, nws = _mfma_ws(B, C, CO, h, w, dev)
                call('smd_conv3x3_mfma_bwd_weight', xp.data_ptr(), g_y.data_ptr(), g_w.data_ptr(), ws.data_ptr(), nws, B, C, CO, h, w, pieces, _stream())
            ok = CO % 32 == 0 or thin
            ref_wgt = (lambda: thin_bwd(False, True)[1]) if (thin and not bf) else (lambda: cb([False, True, False])[1].float())
            if ok and (force or _conv_route('wgt' + sfx, B, C, CO, h, w, run_wgt, ref_wgt)): run_wgt()
            else: g_w = ref_wgt()
        return g_xp, g_w, None, None


def conv3x3_mfma(xp, weight, pieces: int = 3):
    """`F.conv2d(xp, weight)` for an input that is already reflection-padded, ALWAYS through the split-bf16 MFMA kernels (`smd_conv3x3_mfma_*`): the wide
    up-convolutions of the decoder (src/networks/decoders/monodepth.py:40-50, 71-84), bias-free (the next glue kernel adds it).  xp (B,C,h+2,w+2) fp32,
    weight (CO,C,3,3) fp32 -> (B,CO,h,w) fp32; C % 16 == 0 and CO % 32 == 0, or the thin stage CO = 16 with C = 16 | 32 (`_lib.Unsupported` otherwise).  Every fp32 operand is split exactly into three
    bf16 pieces and six products are kept per fp32 product (`pieces=3`: fp32-class error, see csrc/smd_conv_mfma.hip; `pieces=2` is an experiment setting)."""
    return _Conv3x3Wide.apply(xp, weight, int(pieces), True)


def conv3x3_wide(xp, weight):
    """The same convolution, each operator through whichever of the MFMA kernels and MIOpen won this box's A/B for its shape (`_conv_route`)."""
    return _Conv3x3Wide.apply(xp, weight, 3, False)


class _EluUpCatPad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, bias, skip, out_dtype):
        a = _check_fb('a', a)
        if a.ndim != 4: raise ValueError(f'expected (B,C,h,w), got {tuple(a.shape)}')
        B, Ca, h, w = a.shape
        if bias is not None: bias = _check('bias', bias, (Ca,))
        Cs = 0
        if skip is not None:
            Cs = skip.shape[1]
            skip = _check_fb('skip', skip, (B, Cs, 2*h, 2*w))
        out = torch.empty((B, Ca + Cs, 2*h + 2, 2*w + 2), device=a.device, dtype=out_dtype)
        dt = (1 if a.dtype == _BF else 0) | (2 if (skip is not None and skip.dtype == _BF) else 0) | (4 if out_dtype == _BF else 0)
        call('smd_elu_up_cat_pad_fwd', a.data_ptr(), bias.data_ptr() if bias is not None else None, skip.data_ptr() if skip is not None else None,
             out.data_ptr(), B, Ca, Cs, h, w, dt, _stream())
        ctx.save_for_backward(a, bias)
        ctx.Cs, ctx.dt, ctx.out_dtype, ctx.skip_dtype = Cs, dt, out_dtype, (skip.dtype if skip is not None else None)
        return out

    @staticmethod
    def backward(ctx, g_out):
        a, bias = ctx.saved_tensors
        _on(a)
        B, Ca, h, w = a.shape
        Cs = ctx.Cs
        g_b = torch.empty_like(bias) if (bias is not None and ctx.needs_input_grad[1]) else None
        g_a = torch.empty_like(a) if (ctx.needs_input_grad[0] or g_b is not None) else None
        g_skip = torch.empty((B, Cs, 2*h, 2*w), device=a.device, dtype=ctx.skip_dtype) if (Cs and ctx.needs_input_grad[2]) else None
        if g_a is None and g_skip is None: return None, None, None, None
        ws, nbytes = _glue_ws(B, Ca, h, w, a.device) if g_b is not None else (None, 0)
        call('smd_elu_up_cat_pad_bwd', a.data_ptr(), bias.data_ptr() if bias is not None else None, g_out.to(ctx.out_dtype).contiguous().data_ptr(),
             g_a.data_ptr() if g_a is not None else None, g_skip.data_ptr() if g_skip is not None else None,
             g_b.data_ptr() if g_b is not None else None, ws.data_ptr() if ws is not None else None, nbytes, B, Ca, Cs, h, w, ctx.dt, _stream())
        return g_a, g_b, g_skip, None


def elu_up_cat_pad(a, skip=None, bias=None, out_dtype=None):
    """reflect_pad1(cat(nearest_x2(elu(a + bias)), skip)): (B,Ca,h,w) [+ (B,Cs,2h,2w)] -> (B,Ca+Cs,2h+2,2w+2).
    a / skip float32 or bfloat16 (independently); `out_dtype` defaults to a's."""
    return _EluUpCatPad.apply(a, bias, skip, out_dtype or a.dtype)


class _BatchNormAct(torch.autograd.Function):
    """Training-mode BatchNorm2d + optional residual add + optional ReLU (`smd_bn_*`)."""

    @staticmethod
    def forward(ctx, x, residual, weight, bias, running_mean, running_var, momentum, eps, relu):
        x = _check('x', x)
        if x.ndim != 4: raise ValueError(f'expected (N,C,H,W), got {tuple(x.shape)}')
        N, C, H, W = x.shape
        if N*H*W < 2: raise ValueError('Expected more than 1 value per channel when training')   # F.batch_norm's own check
        if residual is not None: residual = _check('residual', residual, x.shape)
        weight = _check('weight', weight, (C,)); bias = _check('bias', bias, (C,))
        y = torch.empty_like(x)
        save = torch.empty((2, C), device=x.device, dtype=torch.float32)
        nbytes = _lib.lib.smd_bn_workspace_bytes(N, C, H*W)
        ws = torch.empty(nbytes, device=x.device, dtype=torch.uint8)
        call('smd_bn_fwd', x.data_ptr(), residual.data_ptr() if residual is not None else None, weight.data_ptr(), bias.data_ptr(),
             running_mean.data_ptr() if running_mean is not None else None, running_var.data_ptr() if running_var is not None else None,
             float(momentum), float(eps), int(relu), y.data_ptr(), save[0].data_ptr(), save[1].data_ptr(), ws.data_ptr(), nbytes, N, C, H*W, _stream())
        ctx.save_for_backward(x, y if relu else None, weight, save)
        ctx.relu, ctx.has_res = bool(relu), residual is not None
        return y

    @staticmethod
    def backward(ctx, g_y):
        x, y, weight, save = ctx.saved_tensors
        _on(x)
        N, C, H, W = x.shape
        g_y = g_y.contiguous()
        g_x = torch.empty_like(x)
        g_res = None
        if ctx.has_res and ctx.needs_input_grad[1]: g_res = torch.empty_like(x) if ctx.relu else g_y   # without ReLU the branch gradient IS g_y
        g_w = torch.empty_like(weight); g_b = torch.empty_like(weight)
        nbytes = _lib.lib.smd_bn_workspace_bytes(N, C, H*W)
        ws = torch.empty(nbytes, device=x.device, dtype=torch.uint8)
        call('smd_bn_bwd', x.data_ptr(), y.data_ptr() if y is not None else None, g_y.data_ptr(), weight.data_ptr(), save[0].data_ptr(),
             save[1].data_ptr(), int(ctx.relu), g_x.data_ptr(), g_res.data_ptr() if (g_res is not None and ctx.relu) else None,
             g_w.data_ptr(), g_b.data_ptr(), ws.data_ptr(), nbytes, N, C, H*W, _stream())
        return g_x, g_res, g_w, g_b, None, None, None, None, None


def batch_norm_act(x, weight, bias, running_mean=None, running_var=None, *, residual=None, momentum: float = 0.1, eps: float = 1e-5, relu: bool = False):
    """relu?(batch_norm_train(x) [+ residual]); running statistics are updated in place like `F.batch_norm(training=True)`."""
    return _BatchNormAct.apply(x, residual, weight, bias, running_mean, running_var, momentum, eps, relu)


class _MaxPool3x3s2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _check('x', x)
        if x.ndim != 4: raise ValueError(f'expected (N,C,H,W), got {tuple(x.shape)}')
        N, C, H, W = x.shape
        Ho, Wo = (H - 1)//2 + 1, (W - 1)//2 + 1
        y = torch.empty((N, C, Ho, Wo), device=x.device, dtype=torch.float32)
        idx = torch.empty((N, C, Ho, Wo), device=x.device, dtype=torch.uint8)
        call('smd_maxpool3x3s2_fwd', x.data_ptr(), y.data_ptr(), idx.data_ptr(), N, C, H, W, _stream())
        ctx.save_for_backward(idx); ctx.shape = (N, C, H, W)
        return y

    @staticmethod
    def backward(ctx, g_y):
        (idx,) = ctx.saved_tensors
        _on(idx)
        N, C, H, W = ctx.shape
        g_x = torch.empty((N, C, H, W), device=idx.device, dtype=torch.float32)
        call('smd_maxpool3x3s2_bwd', g_y.contiguous().data_ptr(), idx.data_ptr(), g_x.data_ptr(), N, C, H, W, _stream())
        return g_x


def max_pool3x3s2(x):
    """`F.max_pool2d(x, 3, 2, 1)` with a one-byte argmax and a gather backward."""
    return _MaxPool3x3s2.apply(x)


class _DwConv7x7(torch.autograd.Function):
    """Depthwise 7x7 convolution, stride 1, padding 3 (`smd_dwconv7x7_*`)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x = _check('x', x)
        if x.ndim != 4: raise ValueError(f'expected (N,C,H,W), got {tuple(x.shape)}')
        N, C, H, W = x.shape
        weight = _check('weight', weight, (C, 1, 7, 7))
        if bias is not None: bias = _check('bias', bias, (C,))
        y = torch.empty_like(x)
        call('smd_dwconv7x7_fwd', x.data_ptr(), weight.data_ptr(), bias.data_ptr() if bias is not None else None, y.data_ptr(), N, C, H, W, 0, _stream())
        ctx.save_for_backward(x, weight); ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, g_y):
        x, weight = ctx.saved_tensors
        _on(x)
        N, C, H, W = x.shape
        g_y = g_y.contiguous()
        g_x = g_w = g_b = None
        if ctx.needs_input_grad[0]:
            g_x = torch.empty_like(x)
            call('smd_dwconv7x7_fwd', g_y.data_ptr(), weight.data_ptr(), None, g_x.data_ptr(), N, C, H, W, 1, _stream())
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            g_w = torch.empty_like(weight)
            g_b = torch.empty((C,), device=x.device, dtype=torch.float32) if ctx.has_bias else None
            nbytes = _lib.lib.smd_dwconv7x7_workspace_bytes(C, H, W)
            ws = torch.empty(nbytes, device=x.device, dtype=torch.uint8)
            call('smd_dwconv7x7_wrw', x.data_ptr(), g_y.data_ptr(), g_w.data_ptr(), g_b.data_ptr() if g_b is not None else None, ws.data_ptr(), nbytes,
                 N, C, H, W, _stream())
        return g_x, g_w, g_b


def dwconv7x7(x, weight, bias=None):
    """`F.conv2d(x, weight (C,1,7,7), bias, padding=3, groups=C)`."""
    return _DwConv7x7.apply(x, weight, bias)


class _LayerNormCF(torch.autograd.Function):
    """LayerNorm over the channel dimension of an NCHW tensor (`smd_layernorm_cf_*`)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, out_bf16):
        x = _check('x', x)
        if x.ndim != 4: raise ValueError(f'expected (N,C,H,W), got {tuple(x.shape)}')
        N, C, H, W = x.shape
        weight = _check('weight', weight, (C,)); bias = _check('bias', bias, (C,))
        y = torch.empty_like(x, dtype=torch.bfloat16 if out_bf16 else torch.float32)
        stats = torch.empty((2, N*H*W), device=x.device, dtype=torch.float32)
        call('smd_layernorm_cf_fwd', x.data_ptr(), weight.data_ptr(), bias.data_ptr(), y.data_ptr(), int(out_bf16), stats[0].data_ptr(),
             stats[1].data_ptr(), N, C, H*W, float(eps), _stream())
        ctx.save_for_backward(x, weight, stats)
        return y

    @staticmethod
    def backward(ctx, g_y):
        x, weight, stats = ctx.saved_tensors
        _on(x)
        N, C, H, W = x.shape
        if g_y.dtype not in (torch.float32, torch.bfloat16): g_y = g_y.float()
        g_y = g_y.contiguous()
        g_x = torch.empty_like(x); g_w = torch.empty_like(weight); g_b = torch.empty_like(weight)
        nbytes = _lib.lib.smd_layernorm_cf_workspace_bytes(N, C, H*W)
        ws = torch.empty(nbytes, device=x.device, dtype=torch.uint8)
        call('smd_layernorm_cf_bwd', x.data_ptr(), g_y.data_ptr(), int(g_y.dtype == torch.bfloat16), weight.data_ptr(), stats[0].data_ptr(),
             stats[1].data_ptr(), g_x.data_ptr(), g_w.data_ptr(), g_b.data_ptr(), ws.data_ptr(), nbytes, N, C, H*W, _stream())
        return g_x, g_w, g_b, None, None


def layer_norm_cf(x, weight, bias, eps: float = 1e-6, out_dtype=torch.float32):
    """`F.layer_norm(x.permute(0,2,3,1), (C,), weight, bias, eps).permute(0,3,1,2)` without the permutes.  `out_dtype=bfloat16`
    writes the result (and reads its gradient) in bf16 for a bf16 consumer; the arithmetic is fp32."""
    if out_dtype not in (torch.float32, torch.bfloat16): raise TypeError(f'out_dtype must be float32 or bfloat16, got {out_dtype}')
    return _LayerNormCF.apply(x, weight, bias, eps, out_dtype == torch.bfloat16)


# ---------------------------------------------------------------------------------------------------
class _PoseMatrices(torch.autograd.Function):
    """`T_from_AAt` (+ rigid inverse where flagged) in one launch (src/tools/geometry.py:181-209, src/core/trainer.py:253)."""

    @staticmethod
    def forward(ctx, aa, t, invert):
        aa = _check('aa', aa); t = _check('t', t, aa.shape)
        if aa.ndim != 2 or aa.shape[1] != 3: raise ValueError(f'aa and t must be (N,3), got {tuple(aa.shape)}')
        N = aa.shape[0]
        if invert is not None:
            if invert.dtype != torch.uint8 or tuple(invert.shape) != (N,) or not invert.is_cuda: raise ValueError('invert must be a CUDA uint8 (N,) tensor')
            invert = invert.contiguous()
        T = torch.empty((N, 4, 4), device=aa.device, dtype=torch.float32)
        call('smd_pose_fwd', aa.data_ptr(), t.data_ptr(), invert.data_ptr() if invert is not None else None, N, T.data_ptr(), _stream())
        ctx.save_for_backward(aa, t, invert)
        return T

    @staticmethod
    def backward(ctx, g_T):
        aa, t, invert = ctx.saved_tensors
        _on(aa)
        g_T = g_T.contiguous()
        g_aa, g_t = torch.empty_like(aa), torch.empty_like(t)
        call('smd_pose_bwd', aa.data_ptr(), t.data_ptr(), invert.data_ptr() if invert is not None else None, aa.shape[0], g_T.data_ptr(),
             g_aa.data_ptr(), g_t.data_ptr(), _stream())
        return g_aa, g_t, None


def pose_matrices(aa, t, invert=None):
    """Axis-angle + translation (N,3) -> (N,4,4) transforms; rows with `invert[i] != 0` hold the inverse transform."""
    return _PoseMatrices.apply(aa, t, invert)


class _Intrinsics(torch.autograd.Function):
    """`resize_K(build_K(fs, cs), (h, w))` and its inverse in one launch (src/networks/pose.py:60-73, geometry.py:249-263, 383)."""

    @staticmethod
    def forward(ctx, fs, cs, h, w):
        fs = _check('fs', fs); cs = _check('cs', cs, fs.shape)
        if fs.ndim != 2 or fs.shape[1] != 2: raise ValueError(f'fs and cs must be (b,2), got {tuple(fs.shape)}')
        b = fs.shape[0]
        K = torch.empty((b, 4, 4), device=fs.device, dtype=torch.float32); K_inv = torch.empty_like(K)
        call('smd_intrinsics_fwd', fs.data_ptr(), cs.data_ptr(), None, b, h, w, K.data_ptr(), K_inv.data_ptr(), _stream())
        ctx.save_for_backward(fs, cs); ctx.size = (h, w)
        return K, K_inv

    @staticmethod
    def backward(ctx, g_K, g_Kinv):
        fs, cs = ctx.saved_tensors
        _on(fs)
        h, w = ctx.size
        g_fs, g_cs = torch.empty_like(fs), torch.empty_like(cs)
        call('smd_intrinsics_bwd', fs.data_ptr(), cs.data_ptr(), fs.shape[0], h, w, g_K.contiguous().data_ptr(), g_Kinv.contiguous().data_ptr(),
             g_fs.data_ptr(), g_cs.data_ptr(), _stream())
        return g_fs, g_cs, None, None


def intrinsics(fs, cs, size):
    """Normalised focal lengths / principal point (b,2) -> (K, K_inv) (b,4,4) at image size `size=(h, w)`."""
    return _Intrinsics.apply(fs, cs, int(size[0]), int(size[1]))


def inv_intrinsics(K):
    """Inverse of caller-supplied intrinsics (b,4,4) (3x3 block; not differentiable — dataset intrinsics are constants)."""
    K = _check('K', K.detach())
    if K.ndim != 3 or tuple(K.shape[1:]) != (4, 4): raise ValueError(f'K must be (b,4,4), got {tuple(K.shape)}')
    K_inv = torch.empty_like(K)
    call('smd_intrinsics_fwd', None, None, K.data_ptr(), K.shape[0], 1, 1, None, K_inv.data_ptr(), _stream())
    return K_inv


def crop_resize(tensors, crop_shape, out_shape, K=None):
    """Centre crop + bilinear resize of every tensor in `tensors` ((..., H, W) float32, same H, W) and of the intrinsics `K`
    (..., 4, 4), in one launch: `crop_aug` + `resize_aug` of src/core/aspect_ratio.py:67-151 without materialising the crop.
    -> ([(..., oh, ow) ...], K' or None).  Not differentiable (the reference runs it under `no_grad`, on the data)."""
    if not 1 <= len(tensors) <= 8: raise ValueError('1 to 8 tensors per call')
    H, W = tensors[0].shape[-2:]
    ch, cw = (int(v) for v in crop_shape); oh, ow = (int(v) for v in out_shape)
    ts = []
    for i, t in enumerate(tensors):
        t = _check(f'tensors[{i}]', t.detach())
        if tuple(t.shape[-2:]) != (H, W): raise ValueError(f'tensors[{i}]: expected (..., {H}, {W}), got {tuple(t.shape)}')
        ts.append(t)
    outs = [torch.empty((*t.shape[:-2], oh, ow), device=t.device, dtype=torch.float32) for t in ts]
    Kc = Ko = None
    if K is not None:
        Kc = _check('K', K.detach())
        if tuple(Kc.shape[-2:]) != (4, 4): raise ValueError(f'K must be (..., 4, 4), got {tuple(K.shape)}')
        Ko = torch.empty_like(Kc)
    call('smd_crop_resize', ptr_array([t.data_ptr() for t in ts]), ptr_array([o.data_ptr() for o in outs]),
         int_array([t.numel()//(H*W) for t in ts]), len(ts), H, W, ch, cw, oh, ow, Kc.data_ptr() if Kc is not None else None,
         Ko.data_ptr() if Ko is not None else None, Kc.numel()//16 if Kc is not None else 0, _stream())
    return outs, Ko


class _Blur3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _check('x', x)
        if x.ndim < 2: raise ValueError(f'gaussian_blur3x3 needs (..., h, w), got {tuple(x.shape)}')
        h, w = x.shape[-2:]
        out = torch.empty_like(x)
        call('smd_gaussian_blur3x3', x.data_ptr(), out.data_ptr(), x.numel()//(h*w), h, w, 0, _stream())
        return out

    @staticmethod
    def backward(ctx, g):
        _on(g)
        g = _check('grad', g)
        h, w = g.shape[-2:]
        gx = torch.empty_like(g)
        call('smd_gaussian_blur3x3', g.data_ptr(), gx.data_ptr(), g.numel()//(h*w), h, w, 1, _stream())
        return gx


def gaussian_blur3x3(x):
    """`kornia.filters.gaussian_blur2d(x, kernel_size=(3, 3), sigma=(1, 1))` (src/regularizers/smooth.py:21) on (..., h, w) float32: separable
    3-tap Gaussian, reflect border; differentiable (the backward is the transposed map).  h, w >= 2."""
    return _Blur3.apply(x)


def disp_smooth_blurred(disps: dict, imgs, *, use_edges: bool = False, want_aux: bool = True):
    """`handlers.disp_smooth` with `SmoothReg(use_blur=True)`, first-order form (src/regularizers/smooth.py:21, 71-97; handlers.py:262-281):
    per scale, the mean-normalised disparity and the resized image are blurred before the absolute differences are taken.

    Built from the launches that exist: the image is resized with `crop_resize` (crop = frame) and blurred; the disparity is blurred and then
    shifted by (mean(disp) - mean(blur(disp))) per sample — the fused sweep normalises its input by that input's own mean, only DIFFERENCES of
    the normalised field enter the loss, and with the shift the mean it divides by is mean(disp), so what it evaluates is
    |d blur(disp / mean(disp))| exactly as the reference orders it (the blur is linear).  -> (loss, disp_grad|None, image_grad|None)."""
    keys = [int(k) for k in disps.keys()]
    total, aux = 0., (None, None)
    H, W = imgs.shape[-2:]
    for i, (k, d) in enumerate(zip(keys, disps.values())):
        hs, ws = d.shape[-2:]
        img_s = imgs if (hs, ws) == (H, W) else crop_resize([imgs], (H, W), (hs, ws))[0][0]
        bd = gaussian_blur3x3(d)
        x = bd + (d.mean(dim=(2, 3), keepdim=True) - bd.mean(dim=(2, 3), keepdim=True))
        l, dg, ig = disp_smooth_fused({k: x}, gaussian_blur3x3(img_s), use_edges=use_edges, want_aux=want_aux and k == 0)
        total = total + l
        if k == 0: aux = (dg, ig)      # the reference returns the maps of scale KEY 0 (`ls[0][1]`, src/core/handlers.py:280), wherever it sits in the dict
    return total/len(keys), aux[0], aux[1]


def lane_shift_selftest(device='cuda'):
    """Returns (left, right): left[l] = l-1 (0 at lane 0), right[l] = l+1 (0 at lane 63) if the DPP wave shifts that the
    stencil kernels rely on behave as documented."""
    left = torch.empty(64, device=device, dtype=torch.float32); right = torch.empty_like(left)
    call('smd_debug_lane_shift', left.data_ptr(), right.data_ptr(), _stream())
    return left, right
