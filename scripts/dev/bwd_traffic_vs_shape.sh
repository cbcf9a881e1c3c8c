#!/bin/bash
# VERDICT r5 item 4, measured with the launch-shape knobs that exist: time (HIP events around the kernel, scripts/dev/microbench.py) and fabric traffic
# (rocprofv3 --pmc FETCH_SIZE, own pass) of the four-support backward at cfg 5 for: the default (one wave per strip, supports in turn, a block = four scales of a strip),
# four waves per strip (supports concurrently: the target-side rows are shared through L1), shorter strips, and a block of four strips of one scale.  (GPU box.)
cd "$GRAFT_REPO_ROOT"
for kn in "" "bwd_wps=4" "bwd_rh=8" "bwd_rh=8,bwd_wps=4" "bwd_scales_block=0" "bwd_rh=12" "bwd_rh=24"; do
  echo "== knobs: ${kn:-default}"
  MB_PATH=node MB_KNOBS=$kn python scripts/dev/microbench.py cfg5 20 2>&1 | tail -1 | cut -c1-260
  MB_PATH=node MB_KNOBS=$kn PMC_TIMEOUT=120 PMC_GROUPS="FETCH_SIZE;WRITE_SIZE" bash scripts/pmc.sh tmp_shape python $GRAFT_REPO_ROOT/scripts/dev/microbench.py cfg5 5 2>&1 | grep -A2 "k_recon_bwd" | head -3
done
