#!/bin/bash
# Forward kernel at 3 / 4 waves per SIMD (GPU box): how much does one more resident wave buy?  usage: scripts/dev/fwd_lb.sh
cd "$GRAFT_REPO_ROOT/slowtv_monodepth_amd/csrc"
cp smd_recon_fwd.hip /tmp/fwd_orig.hip
for lb in 4 3 2; do
  sed "s/(N <= 2 ? 4 : 3)) void k_recon_main/(N <= 2 ? $lb : 3)) void k_recon_main/" /tmp/fwd_orig.hip > smd_recon_fwd.hip
  rm -f smd_recon_fwd.o; make -s >/dev/null 2>&1
  echo -n "waves/SIMD cap $lb: "
  (cd "$GRAFT_REPO_ROOT" && timeout 100 python scripts/dev/microbench.py cfg2 20 2>&1 | tail -1 | cut -c1-110)
done
cp /tmp/fwd_orig.hip smd_recon_fwd.hip; rm -f smd_recon_fwd.o; make -s >/dev/null 2>&1
