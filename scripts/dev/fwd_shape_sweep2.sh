cd "$GRAFT_REPO_ROOT"
run() { echo -n "[$*] "; env "$@" SMD_BWD_SKIP=0 timeout 200 python scripts/dev/microbench.py ${CFG:-cfg2} 20 2>&1 | tail -1 | sed 's/ | entry points.*//' | sed 's/.*\] fwd/fwd/' | cut -c1-60; }
for rep in 1 2; do
for sh in 1 0; do
run SMD_FWD_SHARE=$sh
run SMD_FWD_SHARE=$sh SMD_FWD_RH=16 SMD_FWD_TAPER_B=0
run SMD_FWD_SHARE=$sh SMD_FWD_RH=24 SMD_FWD_TAPER_B=0
run SMD_FWD_SHARE=$sh SMD_FWD_RH=32 SMD_FWD_TAPER_B=0
run SMD_FWD_SHARE=$sh SMD_FWD_RH=48 SMD_FWD_TAPER_B=0
done
done
