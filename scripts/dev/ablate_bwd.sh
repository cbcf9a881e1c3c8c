#!/bin/bash
# Ablation of the fused backward kernel (GPU box): rebuilds the library with parts of k_recon_bwd's memory traffic replaced by
# register arithmetic and times each variant.  usage: scripts/dev/ablate_bwd.sh     (results are NOT numerically meaningful)
cd "$GRAFT_REPO_ROOT/slowtv_monodepth_amd/csrc"
for abl in 0 1 2 4 3 7 8 15; do
  rm -f smd_recon_bwd.o
  make -s EXPERIMENTS=1 EXTRA="-DSMD_ABLATE_BWD=$abl" >/dev/null 2>&1
  for rough in 0 1; do
    echo -n "SMD_ABLATE_BWD=$abl rough=$rough: "
    (cd "$GRAFT_REPO_ROOT" && MB_ROUGH=$rough timeout 100 python scripts/dev/microbench.py cfg2 20 2>&1 | tail -1 | cut -c1-130)
  done
done
rm -f smd_recon_bwd.o; make -s >/dev/null 2>&1
