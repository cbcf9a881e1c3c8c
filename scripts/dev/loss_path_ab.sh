#!/bin/bash
# A/B of the loss path's launch structure on ONE box (GPU box; rocprofv3 kernel trace of scripts/dev/microbench.py, frames prepared ahead):
#   handlers  the separate operators (two autograd nodes + eager glue)         node/guests=1  the single-node loss path (round 5)
#   node/guests=0  the same node with its guest work as launches of their own
# usage: scripts/dev/loss_path_ab.sh [cfg2|cfg4|cfg5]  -> gpurun_out/loss_path_ab_<cfg>.txt
cfg=${1:-cfg2}
out=gpurun_out/loss_path_ab_$cfg.txt; : > $out
run() { tag=$1; shift; echo "== $tag: $*" | tee -a $out
  env "$@" MB_PREP=ahead SMD_BWD_SKIP=0 bash scripts/prof_micro.sh lpab_$tag $cfg > /dev/null 2>&1
  grep -v "^W2026\|^E2026" gpurun_out/prof_lpab_$tag/run.log | tail -1 | cut -c1-400 >> $out
  grep "smd::\|Cijk\|elementwise\|at::native" gpurun_out/prof_lpab_$tag/trace_summary.txt | cut -c1-92,93-150 >> $out; }
run handlers MB_PATH=handlers
run node_guests MB_PATH=node
run node_launches MB_PATH=node MB_KNOBS=loss_path_guests=0
cat $out
