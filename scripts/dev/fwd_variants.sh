#!/bin/bash
# Times the fused forward (scripts/dev/microbench.py, K0 fused and not) under different register caps (GPU box).
cd "$GRAFT_REPO_ROOT/slowtv_monodepth_amd/csrc"
cp smd_recon_fwd.hip /tmp/fwd_orig.hip
for lb in 1 4; do
  sed "s/(N <= 2 ? [0-9] : 3)/(N <= 2 ? $lb : 3)/" /tmp/fwd_orig.hip > smd_recon_fwd.hip
  rm -f smd_recon_fwd.o; make -s >/dev/null 2>&1
  echo -n "min waves/SIMD $lb: "
  (cd "$GRAFT_REPO_ROOT" && for d in 1 0; do MB_DISP=$d timeout 100 python scripts/dev/microbench.py cfg2 20 2>&1 | tail -1 | cut -c1-130; done)
done
cp /tmp/fwd_orig.hip smd_recon_fwd.hip; rm -f smd_recon_fwd.o; make -s >/dev/null 2>&1
