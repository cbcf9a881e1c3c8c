#!/bin/bash
# Times the fused forward built with extra -D flags (GPU box).  usage: scripts/dev/fwd_defs.sh "" "-DSMD_PRIO=1" ...
cd "$GRAFT_REPO_ROOT/slowtv_monodepth_amd/csrc"
for defs in "$@"; do
  rm -f smd_recon_fwd.o; make -s EXPERIMENTS=1 EXTRA="$defs" >/dev/null 2>&1
  for rough in 0 1; do
    echo -n "[$defs] rough=$rough: "
    (cd "$GRAFT_REPO_ROOT" && MB_ROUGH=$rough timeout 100 python scripts/dev/microbench.py cfg2 20 2>&1 | tail -1 | cut -c1-120)
  done
done
rm -f smd_recon_fwd.o; make -s >/dev/null 2>&1
