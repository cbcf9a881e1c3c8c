cd "$GRAFT_REPO_ROOT"
run() { echo -n "[$*] "; env "$@" SMD_BWD_SKIP=0 timeout 200 python scripts/dev/microbench.py ${CFG:-cfg2} 20 2>&1 | tail -1 | sed 's/ | entry points.*//' | sed 's/.*\] fwd/fwd/' | cut -c1-120; }
for rep in 1 2; do
run A=1
run SMD_FWD_RH=28 SMD_FWD_TAPER_B=9 SMD_FWD_TAPER_RH=24
run SMD_FWD_RH=24 SMD_FWD_TAPER_B=0
run SMD_FWD_RH=28 SMD_FWD_TAPER_B=0
run SMD_FWD_RH=32 SMD_FWD_TAPER_B=0
run SMD_FWD_RH=32 SMD_FWD_TAPER_B=6 SMD_FWD_TAPER_RH=24
run SMD_FWD_RH=20 SMD_FWD_TAPER_B=0
run SMD_FWD_RH=16 SMD_FWD_TAPER_B=0
run SMD_FWD_RH=16 SMD_FWD_TAPER_B=4 SMD_FWD_TAPER_RH=12
done
