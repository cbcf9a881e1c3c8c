#!/bin/bash
# Everything profiles/rNN_* is regenerated from, in one GPU call (GPU box): kernel-trace summaries of the bench at cfg 2/4/5, the PMC
# traffic passes, and the one-rank RCCL lines.  usage: scripts/round_profiles.sh r03      (outputs under gpurun_out/)
tag=${1:-r03}
cd "$GRAFT_REPO_ROOT"
for wl in cfg2 cfg4 cfg5; do
  bash scripts/gpu_profile.sh ${tag}_$wl --workload $wl > gpurun_out/prof_${tag}_$wl.log 2>&1
  grep -m1 '"metric"' gpurun_out/prof_${tag}_$wl/bench.log | cut -c1-200
done
PMC_TIMEOUT=400 bash scripts/pmc_traffic.sh $tag cfg2 cfg4 cfg5 > gpurun_out/pmc_traffic_$tag.log 2>&1
tail -5 gpurun_out/pmc_traffic_$tag.log
{
  echo "# bench.py under torch.distributed.run --nproc-per-node 1 with the RCCL process group forced on one rank (SMD_FORCE_DDP=1): the data-parallel wrappers' own cost"
  echo "# command: SMD_FORCE_DDP=1 SMD_DP_IMPL=<impl> python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline"
  for impl in flat ddp; do
    echo "## SMD_DP_IMPL=$impl"
    SMD_FORCE_DDP=1 SMD_DP_IMPL=$impl timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '"metric"'
  done
  echo "## no process group (same box, same run)"
  timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '"metric"'
} > gpurun_out/rccl_one_rank_$tag.txt
cut -c1-160 gpurun_out/rccl_one_rank_$tag.txt
