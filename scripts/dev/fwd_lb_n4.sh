#!/bin/bash
# Forward kernel for three / four supports at 3 vs 2 waves per SIMD (GPU box).  usage: scripts/dev/fwd_lb_n4.sh
cd "$GRAFT_REPO_ROOT/slowtv_monodepth_amd/csrc"
cp smd_recon_fwd.hip /tmp/fwd_orig.hip
for lb in 3 2; do
  sed "s/(N <= 2 ? 4 : 3)) void k_recon_main/(N <= 2 ? 4 : $lb)) void k_recon_main/" /tmp/fwd_orig.hip > smd_recon_fwd.hip
  rm -f smd_recon_fwd.o; make -s >/dev/null 2>&1
  echo -n "N>2 waves/SIMD cap $lb: "
  (cd "$GRAFT_REPO_ROOT" && timeout 200 python scripts/dev/microbench.py cfg5 10 2>&1 | tail -1 | cut -c1-110)
done
cp /tmp/fwd_orig.hip smd_recon_fwd.hip; rm -f smd_recon_fwd.o; make -s >/dev/null 2>&1
