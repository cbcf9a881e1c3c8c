#!/bin/bash
# One-rank RCCL lines (GPU box): what the data-parallel wrappers cost, eager and with the step replayed from HIP graphs.
# usage: scripts/rccl_one_rank.sh r04      -> gpurun_out/rccl_one_rank_r04.txt
tag=${1:-r04}
cd "$GRAFT_REPO_ROOT"
{
  echo "# bench.py under torch.distributed.run --nproc-per-node 1 with the RCCL process group forced on one rank (SMD_FORCE_DDP=1): the data-parallel wrappers' own cost"
  echo "# command: SMD_FORCE_DDP=1 SMD_DP_IMPL=<impl> python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline"
  for impl in flat ddp; do
    echo "## SMD_DP_IMPL=$impl"
    SMD_FORCE_DDP=1 SMD_DP_IMPL=$impl timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '"metric"'
  done
  echo "## no process group (same box, same run)"
  timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '"metric"'
  echo "## the same with the step replayed from HIP graphs (bench.py --graph; config.hip_graph says what was captured): what the host enqueues per step"
  echo "## SMD_DP_IMPL=flat --graph"
  SMD_FORCE_DDP=1 SMD_DP_IMPL=flat timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --graph 2>/dev/null | grep '"metric"'
  echo "## no process group --graph"
  timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --graph 2>/dev/null | grep '"metric"'
} > gpurun_out/rccl_one_rank_$tag.txt
