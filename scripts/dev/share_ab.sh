#!/bin/bash
# Shared target-side LDS ring of the forward (SMD_FWD_SHARE=1) against the per-wave loads (=0), same library (GPU box).
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do for sh in 1 0; do
  for cfg in cfg2 cfg4 cfg5; do for rough in 0 1; do
    [ $rep = 2 ] && [ $cfg != cfg2 ] && continue
    echo -n "[share=$sh $cfg] rough=$rough: "
    SMD_FWD_SHARE=$sh MB_ROUGH=$rough timeout 150 python scripts/dev/microbench.py $cfg 20 2>&1 | tail -1 | cut -c1-150
  done; done
  echo -n "[share=$sh cfg2 cold]: "; SMD_FWD_SHARE=$sh MB_PREP=cold timeout 150 python scripts/dev/microbench.py cfg2 20 2>&1 | tail -1 | cut -c1-150
done; done
