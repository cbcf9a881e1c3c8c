"""world_size-2 `gloo` test of the data-parallel path on CPU (the 8-GPU RCCL run is the driver's): sharded synthetic
batches, DDP gradient averaging with `no_sync()` on accumulation micro-steps, identical replicas after every optimizer step."""
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


def _worker(rank, world, port, accumulate, out_dir):
    import sys
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    from oracle.backend import OracleBackend
    from slowtv_monodepth_amd.synthetic import make_batch
    from slowtv_monodepth_amd.train import StepModule, init_distributed, train_steps, wrap_ddp
    from slowtv_monodepth_amd.trainer import MonoDepthModule
    r, _, w = init_distributed(backend='gloo')
    assert (r, w) == (rank, world)
    cfg = {'net': {'depth': {'enc_name': 'resnet18', 'pretrained': False}, 'pose': {'enc_name': 'resnet18'}},
           'loss': {'img_recon': {'weight': 1, 'use_min': True, 'use_automask': True}, 'disp_smooth': {'weight': 0.001, 'use_edges': True}},
           'optimizer': {'type': 'adamw', 'lr': 1e-3, 'weight_decay': 1e-3}, 'trainer': {'min_depth': 0.1, 'max_depth': 100}}
    torch.manual_seed(0)   # same initial replica on every rank (DDP would broadcast rank 0's anyway)
    module = MonoDepthModule(copy.deepcopy(cfg), loss_backend=OracleBackend())
    opt = module.configure_optimizers()['optimizer']
    model = wrap_ddp(StepModule(module), torch.device('cpu'))
    assert isinstance(model, torch.nn.parallel.DistributedDataParallel)
    batches = [make_batch(1, 64, 96, (-1, 1), seed=100*rank + k) for k in range(2*accumulate)]   # a different shard per rank
    losses = train_steps(model, opt, lambda it: batches[it], len(batches), accumulate=accumulate)
    vec = torch.cat([p.detach().flatten() for p in module.nets.parameters()])
    gathered = [torch.empty_like(vec) for _ in range(world)]
    dist.all_gather(gathered, vec)
    torch.save({'params_equal': all(torch.equal(gathered[0], g) for g in gathered), 'moved': float((vec - vec.mean()).abs().sum()),
                'losses': [l.item() for l in losses]}, os.path.join(out_dir, f'rank{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('accumulate', [1, 2])
def test_ddp_replicas_stay_identical(tmp_path, accumulate):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, accumulate, str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(tmp_path/f'rank{r}.pt') for r in range(world)]
    assert all(r['params_equal'] for r in res), 'replicas diverged: gradients were not averaged identically'
    assert res[0]['losses'] != res[1]['losses'], 'ranks must see different shards'
    assert all(all(l == l for l in r['losses']) for r in res)
