#!/bin/bash
# Times the fused backward under different register caps (GPU box).  usage: scripts/dev/bwd_variants.sh
cd "$GRAFT_REPO_ROOT/slowtv_monodepth_amd/csrc"
cp smd_recon_bwd.hip /tmp/bwd_orig.hip
for lb in ${LBS:-0 4 5}; do
  if [ $lb -gt 0 ]; then sed "s/__launch_bounds__(64\*kWavesPerBlock, 4) void k_recon_bwd/__launch_bounds__(64*kWavesPerBlock, $lb) void k_recon_bwd/" /tmp/bwd_orig.hip > smd_recon_bwd.hip; fi
  rm -f smd_recon_bwd.o; make -s >/dev/null 2>&1
  echo -n "min waves/SIMD $lb: "
  (cd "$GRAFT_REPO_ROOT" && timeout 100 python scripts/dev/microbench.py cfg2 20 2>&1 | tail -1 | cut -c1-110; MB_ROUGH=1 timeout 100 python scripts/dev/microbench.py cfg2 20 2>&1 | tail -1 | cut -c1-110)
done
cp /tmp/bwd_orig.hip smd_recon_bwd.hip; rm -f smd_recon_bwd.o; make -s >/dev/null 2>&1
