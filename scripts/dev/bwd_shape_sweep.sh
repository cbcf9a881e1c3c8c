#!/bin/bash
# Launch shapes of the fused backward on one box (GPU box): strip height and taper through the library's environment switches.
cd "$GRAFT_REPO_ROOT"
run() { echo -n "[$*] "; env "$@" SMD_BWD_SKIP=0 timeout 200 python scripts/dev/microbench.py ${CFG:-cfg2} 20 2>&1 | tail -1 | sed 's/ | entry points.*//' | sed 's/.*| bwd/bwd/' | cut -c1-60; }
for rep in 1 2; do
run A=1
run SMD_BWD_RH=12
run SMD_BWD_RH=8
run SMD_BWD_RH=16 SMD_BWD_TAPER_B=0
run SMD_BWD_RH=12 SMD_BWD_TAPER_B=0
run SMD_BWD_RH=16 SMD_BWD_TAPER_B=4
run SMD_BWD_RH=16 SMD_BWD_TAPER_B=2 SMD_BWD_TAPER_RH=12
run SMD_BWD_RH=16 SMD_BWD_TAPER_B=3 SMD_BWD_TAPER_RH=4
done
