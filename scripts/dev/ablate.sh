#!/bin/bash
# Ablation of the fused forward kernel (GPU box): rebuilds the library with parts of k_recon_main's memory traffic replaced by
# register arithmetic and times each variant.  usage: scripts/dev/ablate.sh [cfg2]     (results are NOT numerically meaningful)
cfg=${1:-cfg2}
cd "$GRAFT_REPO_ROOT/slowtv_monodepth_amd/csrc"
for abl in ${ABLS:-0 1 2 3 7}; do
  rm -f smd_recon_fwd.o
  make -s EXPERIMENTS=1 EXTRA="-DSMD_ABLATE=$abl" >/dev/null 2>&1
  echo -n "SMD_ABLATE=$abl: "
  (cd "$GRAFT_REPO_ROOT" && timeout 100 python scripts/dev/microbench.py $cfg 20 2>&1 | tail -1 | cut -c1-110)
done
rm -f smd_recon_fwd.o; make -s >/dev/null 2>&1
