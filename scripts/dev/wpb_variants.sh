#!/bin/bash
# Times the fused kernels with different numbers of (independent) waves per workgroup (GPU box).  usage: scripts/dev/wpb_variants.sh
cd "$GRAFT_REPO_ROOT/slowtv_monodepth_amd/csrc"
for wpb in 1 2 4 8; do
  rm -f smd_recon_fwd.o smd_recon_bwd.o smd_api.o; make -s EXPERIMENTS=1 EXTRA="-DSMD_WAVES_PER_BLOCK=$wpb" >/dev/null 2>&1
  echo -n "waves/block $wpb: "
  (cd "$GRAFT_REPO_ROOT" && timeout 100 python scripts/dev/microbench.py cfg2 20 2>&1 | tail -1 | cut -c1-120)
done
rm -f smd_recon_fwd.o smd_recon_bwd.o smd_api.o; make -s >/dev/null 2>&1
