"""Training entry point: `python -m slowtv_monodepth_amd.train -c cfg.yaml [override.yaml ...] -n NAME [-v 0] [-s 42] [-g N]`.

Mirrors the flags and cfg handling of the reference's `api/train/train.py:16-33` (ordered YAML merge, seed, number of
GPUs) with an own loop instead of PyTorch-Lightning: one process per GPU (`torchrun`/`torch.distributed.run` sets
RANK/LOCAL_RANK/WORLD_SIZE), gradients averaged over RCCL by a few flat-bucket all-reduces (`FlatAllReduce`; stock DDP with
overlap is selectable), no collective on gradient-accumulation micro-steps (`trainer.accumulate_grad_batches`, train.py:110), AdamW +
StepLR∘LinearLR.  Data are device-resident synthetic triplets (`synthetic.make_batch`): datasets are out of scope here
(SURVEY.md §2), so only `supp_idxs` is read from a cfg's `dataset` section and a cfg that names a real dataset `type` is
refused unless `--synthetic-data` says that synthetic triplets of its shape are intended.  `last.ckpt` is written in the
reference's Lightning layout (`state_dict` with the reference's parameter names; `networks/checkpoint.py`).
"""
from __future__ import annotations

import argparse
import os
import time
from contextlib import nullcontext
from pathlib import Path

import torch
import torch.distributed as dist
import torch.nn as nn

from . import io
from .synthetic import make_batch
from .trainer import MonoDepthModule

__all__ = ['StepModule', 'FlatAllReduce', 'wrap_ddp', 'train_steps', 'init_distributed', 'main']


class StepModule(nn.Module):
    """DDP hooks the module whose `forward` it is given; the training step is that forward (as Lightning does)."""
    def __init__(self, module: MonoDepthModule):
        super().__init__()
        self.module = module

    def forward(self, batch):
        loss, loss_dict, _ = self.module.step(batch)
        return loss, {k: v.detach() for k, v in loss_dict.items() if k.startswith('loss_')}


def init_distributed(backend: str | None = None) -> tuple[int, int, int]:
    """(rank, local_rank, world) from the torchrun environment; initialises the process group when world > 1."""
    world = int(os.environ.get('WORLD_SIZE', 1)); rank = int(os.environ.get('RANK', 0)); local = int(os.environ.get('LOCAL_RANK', 0))
    force = os.environ.get('SMD_FORCE_DDP') == '1' and 'MASTER_ADDR' in os.environ   # exercise RCCL + DDP on a single GPU
    if (world > 1 or force) and not dist.is_initialized():
        backend = backend or ('nccl' if torch.cuda.is_available() else 'gloo')  # 'nccl' is RCCL on ROCm
        if backend == 'nccl': torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


class FlatAllReduce(nn.Module):
    """Data-parallel wrapper without per-parameter device work: gradients are gathered into a few flat buckets (one
    `torch.cat(..., out=bucket)` launch per bucket) and every bucket is averaged across ranks by ONE RCCL all-reduce, issued
    asynchronously the moment the last gradient of the bucket has been accumulated, so the collective runs on RCCL's stream
    over xGMI while backward continues; after backward each `p.grad` is simply re-pointed at its slice of the averaged bucket
    (no copy back: the optimizer reads the bucket).  Public PyTorch API only.

    Measured against `DistributedDataParallel` on this network (140 parameter tensors, 108 MB of gradients, 17 ms step) with
    the process group forced on one rank: DDP's reducer launches one scale-and-copy kernel per parameter inside backward and,
    with the two networks on two streams, serialises them: +2.6 ms per step; this wrapper: +0.25 ms.
    `SMD_DP_IMPL=ddp` selects the stock wrapper, `SMD_DP_OVERLAP=0` defers every all-reduce to the end of backward.
    """
    def __init__(self, step: nn.Module, bucket_cap_mb: int = 32, overlap: bool | None = None):
        super().__init__()
        self.module = step
        self.world = dist.get_world_size()
        self.avg_op = dist.ReduceOp.AVG if dist.get_backend() == 'nccl' else dist.ReduceOp.SUM   # gloo has no AVG
        self.overlap = (os.environ.get('SMD_DP_OVERLAP', '1') != '0') if overlap is None else overlap
        self.require_sync = True     # set False on gradient-accumulation micro-steps (no collective)
        with torch.no_grad():        # replicas start identical (what DDP's constructor does)
            for t in list(step.parameters()) + list(step.buffers()): dist.broadcast(t, 0)
        named = [(n, p) for n, p in step.named_parameters() if p.requires_grad]
        params = [p for _, p in named]
        # Buckets never mix networks: autograd accumulates a network's gradients on the stream its forward ran on (the pose
        # network lives on a side stream), so a bucket packed on "the current stream" of its last gradient is ordered after
        # all of its gradients without any cross-stream wait.
        groups = {}
        for n, p in named: groups.setdefault(n.split('nets.')[-1].split('.')[0] if 'nets.' in n else '', []).append(p)
        self.buckets, cap = [], bucket_cap_mb*(1 << 20)//4
        for ps in groups.values():
            cur, cur_n = [], 0
            for p in reversed(ps):   # roughly the order in which backward finishes them
                cur.append(p); cur_n += p.numel()
                if cur_n >= cap: self.buckets.append(cur); cur, cur_n = [], 0
            if cur: self.buckets.append(cur)
        self.flats = [torch.zeros(sum(p.numel() for p in bk), device=bk[0].device, dtype=bk[0].dtype) for bk in self.buckets]
        self.views = [[v.view_as(p) for v, p in zip(flat.split([p.numel() for p in bk]), bk)] for flat, bk in zip(self.flats, self.buckets)]
        self._bucket_of = {p: i for i, bk in enumerate(self.buckets) for p in bk}
        self._left = [len(bk) for bk in self.buckets]
        self._works = [None]*len(self.buckets)
        self._streams = [set() for _ in self.buckets]   # streams on which the gradients of a bucket were accumulated
        # Collectives must be issued in the same order on every rank: a completed bucket is only launched once all of its
        # predecessors in `_order` have been.  ANY order shared by the ranks is correct; a poor one only delays launches.  The
        # first synchronised step therefore already overlaps, with the construction order (per network, reverse parameter order =
        # roughly the order in which backward finishes them); it records the order in which the buckets really completed, rank 0's
        # record is broadcast, and later steps use that.
        self._order, self._order_final, self._arrival, self._next, self._ready = list(range(len(self.buckets))), False, [], 0, set()
        if self.overlap:
            for p in params: p.register_post_accumulate_grad_hook(self._on_grad)

    def forward(self, *args, **kwargs): return self.module(*args, **kwargs)
    # NOTE for callers: after `sync_gradients()` every `p.grad` is a view of a bucket; clear gradients with
    # `optimizer.zero_grad(set_to_none=True)` (as `train_steps` does) so that the next backward produces fresh tensors to pack.

    def _on_grad(self, p) -> None:
        if not self.require_sync: return
        i = self._bucket_of[p]
        if p.is_cuda: self._streams[i].add(torch.cuda.current_stream(p.device))
        self._left[i] -= 1
        if self._left[i] == 0:
            self._arrival.append(i)
            self._ready.add(i)
            while self._next < len(self._order) and self._order[self._next] in self._ready:
                self._launch(self._order[self._next], in_backward=True)
                self._next += 1

    @torch.no_grad()
    def _launch(self, i: int, in_backward: bool) -> None:
        bk = self.buckets[i]
        for p in bk:
            if p.grad is None: p.grad = torch.zeros_like(p)
        if in_backward and bk[0].is_cuda:   # launched from another stream's hook (deferred) or mixed bucket: order the pack after its producers
            cur = torch.cuda.current_stream(bk[0].device)
            for st in self._streams[i]:
                if st != cur: cur.wait_stream(st)
        # pack: one launch per bucket.  A gradient that already IS its slice of the bucket (the caller kept the views of the last step:
        # `zero_grad(set_to_none=False)`, or a second backward accumulated into them) must not be an input of a `cat` whose output it aliases.
        alias = [p.grad.data_ptr() == v.data_ptr() for p, v in zip(bk, self.views[i])]
        if not any(alias): torch.cat([p.grad.reshape(-1) for p in bk], out=self.flats[i])
        elif not all(alias):
            for p, v, a in zip(bk, self.views[i], alias):
                if not a: v.copy_(p.grad)
        self._works[i] = dist.all_reduce(self.flats[i], op=self.avg_op, async_op=True)

    @torch.no_grad()
    def sync_gradients(self) -> None:
        """Finish the gradient average (call after the last backward of an optimizer step, before the optimizer)."""
        for i in self._order:
            if self._works[i] is None: self._launch(i, in_backward=False)
        if not self._order_final and self.overlap:
            seen = list(dict.fromkeys(self._arrival))
            seen += [i for i in range(len(self.buckets)) if i not in seen]
            t = torch.tensor(seen, dtype=torch.int64, device=self.flats[0].device)
            dist.broadcast(t, 0)
            self._order, self._order_final = [int(v) for v in t.tolist()], True
        self._arrival, self._next = [], 0
        self._ready.clear()
        for i, bk in enumerate(self.buckets):
            self._works[i].wait()
            if self.avg_op == dist.ReduceOp.SUM: self.flats[i].div_(self.world)
            for p, v in zip(bk, self.views[i]): p.grad = v      # unpack without a copy: the optimizer reads the averaged bucket
            self._works[i] = None
            self._left[i] = len(bk)
            self._streams[i].clear()


    @torch.no_grad()
    def average_static(self, grads: list) -> None:
        """Average gradients that live in FIXED tensors (`grads[i][k]` belongs to `self.buckets[i][k]`): pack, all-reduce, wait — nothing else.
        For a training step replayed from HIP graphs (`bench.py --graph`): forward + backward are one graph that rewrites the same gradient
        tensors on every replay, the optimizer step is a second graph that reads the bucket views, and the collectives run here, eagerly,
        between the two.  (A collective INSIDE a capture is at the mercy of the process group's watchdog thread, which polls events while
        the capture is open: one run in two to four died on this stack.)  Hooks must be inert: set `require_sync = False`."""
        works = []
        for i, gs in enumerate(grads):
            torch.cat([g.reshape(-1) for g in gs], out=self.flats[i])
            works.append(dist.all_reduce(self.flats[i], op=self.avg_op, async_op=True))
        for i, w in enumerate(works):
            w.wait()
            if self.avg_op == dist.ReduceOp.SUM: self.flats[i].div_(self.world)


def wrap_ddp(step: StepModule, device: torch.device, bucket_cap_mb: int = 25) -> nn.Module:
    """Data parallelism over RCCL/xGMI, one process per GPU.  Default: `FlatAllReduce` (see there).  `SMD_DP_IMPL=ddp`:
    torch DDP with `bucket_cap_mb` buckets reduced as backward produces them, buckets aliasing the .grad tensors.
    BatchNorm statistics stay per rank either way (the reference does not enable SyncBN)."""
    force = os.environ.get('SMD_FORCE_DDP') == '1'
    if not (dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force)): return step
    if os.environ.get('SMD_DP_IMPL', 'flat') != 'ddp': return FlatAllReduce(step)
    ids = [device.index] if device.type == 'cuda' else None
    return nn.parallel.DistributedDataParallel(step, device_ids=ids, broadcast_buffers=False, gradient_as_bucket_view=True,
                                               bucket_cap_mb=bucket_cap_mb)  # (static_graph would forbid no_sync() on the first micro-step)


_SEEDS: dict = {}


def _grad_seed(loss: torch.Tensor, accumulate: int) -> torch.Tensor:
    """The gradient `(loss/accumulate).backward()` would start from — fl(1/accumulate) in the loss's dtype, the quotient the division's own backward
    forms — as a cached tensor: the division, the `ones_like` fill and the division's backward are three latency-sized launches that sit between the
    loss path's forward and backward on every step."""
    key = (loss.device, loss.dtype, tuple(loss.shape), int(accumulate))
    seed = _SEEDS.get(key)
    if seed is None:
        seed = _SEEDS[key] = torch.ones(loss.shape, device=loss.device, dtype=loss.dtype)/accumulate
    return seed


def train_steps(model: nn.Module, opt: torch.optim.Optimizer, batch_fn, steps: int, accumulate: int = 1, clip=None, detect_anomaly: bool = False):
    """Run `steps` optimizer micro-steps (each = forward + backward on one batch; the optimizer fires every `accumulate`).
    LR schedulers step per epoch in the reference (Lightning's default interval), so they are the caller's business.
    Returns the list of (detached) loss tensors; nothing in here synchronises with the host unless `detect_anomaly` is set, which
    mirrors the reference's `DetectAnomaly` callback (src/utils/callbacks.py:27-31): raise on a non-finite loss, every batch."""
    losses = []
    ddp = isinstance(model, nn.parallel.DistributedDataParallel)
    flat = isinstance(model, FlatAllReduce)
    for it in range(steps):
        boundary = (it + 1) % accumulate == 0
        ctx = model.no_sync() if (ddp and not boundary) else nullcontext()
        if flat: model.require_sync = boundary
        with ctx:
            loss, _ = model(batch_fn(it))
            if detect_anomaly and not torch.isfinite(loss).item(): raise ValueError(f'Detected NaN/Infinite loss: "{loss.item()}"')
            loss.backward(gradient=_grad_seed(loss, accumulate))    # == (loss/accumulate).backward(), without its three launches per step
        if boundary:
            if flat: model.sync_gradients()
            if clip: torch.nn.utils.clip_grad_norm_(model.parameters(), clip)
            opt.step()
            opt.zero_grad(set_to_none=True)
        losses.append(loss.detach())
    return losses


def dataset_types(cfg: dict) -> list:
    """Dataset types a cfg names.  The reference keys the `dataset` section BY TYPE — `dataset: {kitti_lmdb: {split: ...}}`,
    `parsers.get_ds` iterates `for t, kw in cfg.items()` (src/tools/parsers.py:109-135) — so every key with a (possibly empty)
    mapping as value is a type; an explicit `type:` field inside an entry is honoured as well."""
    out = set()
    for k, d in (cfg.get('dataset') or {}).items():
        if d is None: continue                      # `key: null` removes an inherited dataset in the reference's cfgs
        if isinstance(d, dict):
            out.add(str(d['type']) if d.get('type') else str(k))
    return sorted(out)


def dataset_supp_idxs(cfg: dict) -> list:
    for d in (cfg.get('dataset') or {}).values():
        if isinstance(d, dict):
            for v in (d, *(x for x in d.values() if isinstance(x, dict))):     # entry-level or per-mode (`train:` / `val:`) override
                if v.get('supp_idxs'): return list(v['supp_idxs'])
    return [-1, 1]


def restore_training_state(ckpt: dict, opt, sched, verbose: bool = True) -> int:
    """`--resume`: optimizer + scheduler state and the epoch to continue from.  -> first epoch to run.

    Optimizer and scheduler go together: either BOTH states are restored, or neither — then the moments start from zero and the fresh scheduler is
    stepped forward to the resumed epoch, so that the learning rate is the one the schedule prescribes there (a StepLR past its decay point would
    otherwise train at the initial rate while the epoch counter says otherwise).  A reference checkpoint carries timm's parameter-group layout,
    which this package's optimizer refuses: that is the case this serves."""
    import copy
    first_epoch = int(ckpt.get('epoch', -1)) + 1
    opt_backup = copy.deepcopy(opt.state_dict())
    sched_backup = copy.deepcopy(sched.state_dict()) if sched is not None else None
    try:
        if not ckpt.get('optimizer_states'): raise KeyError('no optimizer state in the checkpoint')
        opt.load_state_dict(ckpt['optimizer_states'][0])
        if sched is not None:
            if not ckpt.get('lr_schedulers'): raise KeyError('no scheduler state in the checkpoint')
            sched.load_state_dict(ckpt['lr_schedulers'][0])
    except (ValueError, KeyError) as e:
        opt.load_state_dict(opt_backup)
        if sched is not None:
            sched.load_state_dict(sched_backup)
            for _ in range(first_epoch): sched.step()
        if verbose:
            lrs = [round(g['lr'], 10) for g in opt.param_groups]
            print(f'--resume: optimizer / scheduler state not restored ({e}); weights only — optimizer moments start from zero, the scheduler was '
                  f'stepped to epoch {first_epoch} (learning rates {lrs})', flush=True)
    return first_epoch


def main(argv=None):
    p = argparse.ArgumentParser(description='Monocular depth trainer (MI355X hot path).')
    p.add_argument('--cfg-files', '-c', type=Path, nargs='*', required=True, help='YAML configs (default, override...).')
    p.add_argument('--ckpt-dir', '-o', default=Path('runs'), type=Path)
    p.add_argument('--synthetic-data', action='store_true', help='run a cfg that names a real dataset on synthetic triplets of its shape')
    p.add_argument('--name', '-n', required=True, type=str)
    p.add_argument('--version', '-v', default=0, type=int)
    p.add_argument('--seed', '-s', default=42, type=int)
    p.add_argument('--gpus', '-g', default=1, type=int, help='informational: launch with torch.distributed.run --nproc-per-node N')
    p.add_argument('--steps', default=100, type=int, help='optimizer micro-steps per epoch on synthetic data')
    p.add_argument('--shape', default=[192, 640], type=int, nargs=2)
    p.add_argument('--resume', type=Path, default=None, help='checkpoint to continue from (this package\'s last.ckpt or a reference checkpoint): weights, '
                                                               'optimizer and scheduler state, epoch counter')
    args = p.parse_args(argv)

    cfg = io.load_merge_yaml(*args.cfg_files)
    rank, local, world = init_distributed()
    device = torch.device('cuda', local) if torch.cuda.is_available() else torch.device('cpu')
    torch.manual_seed(args.seed)
    module = MonoDepthModule(cfg).to(device)
    conf = module.configure_optimizers()
    opt, sched = conf['optimizer'], conf.get('lr_scheduler')
    tcfg = cfg.get('trainer', {})
    acc = tcfg.get('accumulate_grad_batches', 1)
    if module.auto_scale_lr:
        for g in opt.param_groups: g['lr'] *= world*acc
    b = cfg.get('loader', {}).get('batch_size', 12)
    named = dataset_types(cfg)
    if named and not args.synthetic_data:
        raise SystemExit(f'the cfg names dataset type(s) {named}: this package trains on synthetic triplets only (datasets are out of scope); '
                         'pass --synthetic-data to run the cfg on synthetic triplets of its shape, or drop the dataset `type`')
    supp_idxs = dataset_supp_idxs(cfg)
    batch = make_batch(b, args.shape[0], args.shape[1], supp_idxs, seed=args.seed + rank, device=device)
    model = wrap_ddp(StepModule(module), device)
    save_dir = args.ckpt_dir/args.name/f'{args.version:03}'
    if rank == 0: save_dir.mkdir(parents=True, exist_ok=True)
    first_epoch = 0
    if args.resume is not None:
        from .networks.checkpoint import load_reference_checkpoint
        import pickle
        try: ckpt = torch.load(args.resume, map_location='cpu', weights_only=True)
        except pickle.UnpicklingError:   # a Lightning checkpoint pickles hyper-parameter objects the allow-list refuses: the full unpickler, for a file the user named
            ckpt = torch.load(args.resume, map_location='cpu', weights_only=False)   # (a corrupt / missing file raises something else and is not retried)
        load_reference_checkpoint(module, ckpt)
        first_epoch = restore_training_state(ckpt, opt, sched, verbose=rank == 0)
        if rank == 0: print(f'resumed from {args.resume}: epoch {first_epoch}, global step {ckpt.get("global_step", 0)}', flush=True)
    for epoch in range(first_epoch, tcfg.get('max_epochs', 1)):
        t0 = time.time()
        # (shallow copies: the aspect-ratio augmentation replaces entries of the batch dicts in place, as in the reference)
        losses = train_steps(model, opt, lambda it: tuple(dict(d) for d in batch), args.steps, accumulate=acc, clip=tcfg.get('gradient_clip_val'),
                             detect_anomaly=bool(tcfg.get('detect_anomaly', False)))
        if sched is not None: sched.step()
        last = losses[-1].item()
        if rank == 0:
            dt = time.time() - t0
            print(f'epoch {epoch}: loss {last:.6f}  {args.steps*b*world/dt:.1f} img/s', flush=True)
            from .networks.checkpoint import reference_checkpoint
            torch.save(reference_checkpoint(module, epoch=epoch, global_step=(epoch + 1)*args.steps, optimizer=opt, scheduler=sched), save_dir/'last.ckpt')
    if world > 1: dist.destroy_process_group()


if __name__ == '__main__':
    main()
