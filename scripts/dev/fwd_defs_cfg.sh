#!/bin/bash
# Times the fused forward built with extra -D flags at a given workload (GPU box).  usage: scripts/dev/fwd_defs_cfg.sh cfg5 "" "-DSMD_FWD_CAM_REGS" ...
cfg=$1; shift
cd "$GRAFT_REPO_ROOT/slowtv_monodepth_amd/csrc"
for defs in "$@"; do
  rm -f smd_recon_fwd.o; make -s EXPERIMENTS=1 EXTRA="$defs" >/dev/null 2>&1
  for rough in 0 1; do
    echo -n "[$cfg $defs] rough=$rough: "
    (cd "$GRAFT_REPO_ROOT" && MB_ROUGH=$rough timeout 150 python scripts/dev/microbench.py $cfg 10 2>&1 | tail -1 | cut -c1-160)
  done
done
rm -f smd_recon_fwd.o; make -s >/dev/null 2>&1
