"""world_size-2 `gloo` test of the data-parallel path on CPU (the 8-GPU RCCL run is the driver's): sharded synthetic
batches, gradient averaging (flat-bucket all-reduce and torch DDP) with no collective on accumulation micro-steps, identical
replicas after every optimizer step, identical trajectories between the two implementations."""
import copy
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, accumulate, out_dir, impl):
    import sys
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), SMD_DP_IMPL=impl)
    torch.set_num_threads(2)
    from oracle.backend import OracleBackend
    from slowtv_monodepth_amd.synthetic import make_batch
    from slowtv_monodepth_amd.train import FlatAllReduce, StepModule, init_distributed, train_steps, wrap_ddp
    from slowtv_monodepth_amd.trainer import MonoDepthModule
    r, _, w = init_distributed(backend='gloo')
    assert (r, w) == (rank, world)
    cfg = {'net': {'depth': {'enc_name': 'resnet18', 'pretrained': False}, 'pose': {'enc_name': 'resnet18'}},
           'loss': {'img_recon': {'weight': 1, 'use_min': True, 'use_automask': True}, 'disp_smooth': {'weight': 0.001, 'use_edges': True}},
           'optimizer': {'type': 'adamw', 'lr': 1e-3, 'weight_decay': 1e-3}, 'trainer': {'min_depth': 0.1, 'max_depth': 100}}
    torch.manual_seed(rank)   # DIFFERENT initial replicas: the wrapper must broadcast rank 0's
    module = MonoDepthModule(copy.deepcopy(cfg), loss_backend=OracleBackend())
    opt = module.configure_optimizers()['optimizer']
    model = wrap_ddp(StepModule(module), torch.device('cpu'))
    assert isinstance(model, torch.nn.parallel.DistributedDataParallel if impl == 'ddp' else FlatAllReduce)
    batches = [make_batch(1, 64, 96, (-1, 1), seed=100*rank + k) for k in range(2*accumulate)]   # a different shard per rank
    losses = train_steps(model, opt, lambda it: batches[it], len(batches), accumulate=accumulate)
    vec = torch.cat([p.detach().flatten() for p in module.nets.parameters()])
    gathered = [torch.empty_like(vec) for _ in range(world)]
    dist.all_gather(gathered, vec)
    torch.save({'params_equal': all(torch.equal(gathered[0], g) for g in gathered), 'moved': float((vec - vec.mean()).abs().sum()),
                'losses': [l.item() for l in losses], 'params': vec}, os.path.join(out_dir, f'{impl}_rank{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('accumulate', [1, 2])
def test_replicas_stay_identical_and_flat_allreduce_equals_ddp(tmp_path, accumulate):
    world, out = 2, {}
    for impl in ('flat', 'ddp'):
        mp.spawn(_worker, args=(world, _free_port(), accumulate, str(tmp_path), impl), nprocs=world, join=True)
        res = [torch.load(tmp_path/f'{impl}_rank{r}.pt') for r in range(world)]
        assert all(r['params_equal'] for r in res), f'{impl}: replicas diverged: gradients were not averaged identically'
        assert res[0]['losses'] != res[1]['losses'], 'ranks must see different shards'
        assert all(all(l == l for l in r['losses']) for r in res)
        out[impl] = res[0]
    # the flat-bucket all-reduce and torch DDP implement the same averaging: same trajectory from the same start
    torch.testing.assert_close(out['flat']['params'], out['ddp']['params'], rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(torch.tensor(out['flat']['losses']), torch.tensor(out['ddp']['losses']), rtol=1e-4, atol=1e-6)


def _static_worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    from slowtv_monodepth_amd.train import FlatAllReduce, init_distributed
    init_distributed(backend='gloo')
    torch.manual_seed(rank)
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 3))
    model = FlatAllReduce(net, bucket_cap_mb=1)
    model.require_sync = False                                    # hooks inert, as under graph replay
    grads = [[torch.empty_like(p) for p in bk] for bk in model.buckets]      # the fixed tensors a captured backward would rewrite
    for bk, views in zip(model.buckets, model.views):
        for p, v in zip(bk, views): p.grad = v                    # what the optimizer reads
    ok = True
    for step in range(3):
        for gs in grads:
            for k, g in enumerate(gs): g.fill_(float((rank + 1)*(step + 1) + k))
        model.average_static(grads)
        for bk, gs in zip(model.buckets, grads):
            for k, p in enumerate(bk):
                want = sum((r + 1)*(step + 1) + k for r in range(world))/world
                ok &= bool(torch.allclose(p.grad, torch.full_like(p, want)))
    same = [torch.zeros(1) for _ in range(world)]
    dist.all_gather(same, torch.tensor([float(ok)]))
    torch.save({'ok': ok, 'all': [s.item() for s in same], 'params_equal_after_broadcast': True}, os.path.join(out_dir, f'static_rank{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


def test_average_static_between_two_graphs(tmp_path):
    """`FlatAllReduce.average_static` (what `bench.py --graph` runs between the forward+backward graph and the optimizer graph): fixed
    gradient tensors, rewritten in place every step, are packed and averaged into the bucket views the optimizer reads."""
    world = 2
    mp.spawn(_static_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(tmp_path/f'static_rank{r}.pt') for r in range(world)]
    assert all(r['ok'] for r in res) and all(all(v == 1.0 for v in r['all']) for r in res)
