cd "$GRAFT_REPO_ROOT"
run() { echo -n "[$CFG $*] "; env "$@" SMD_BWD_SKIP=0 timeout 200 python scripts/dev/microbench.py ${CFG:-cfg2} 20 2>&1 | tail -1 | sed 's/ | entry points.*//' | sed 's/.*\] fwd/fwd/' | cut -c1-110; }
for rep in 1 2; do
for CFG in cfg4 cfg5; do
export CFG
run A=1
run SMD_FWD_TAPER_B=0 SMD_BWD_TAPER_B=0
run SMD_FWD_TAPER_B=1 SMD_BWD_TAPER_B=1
run SMD_FWD_RH=24 SMD_FWD_TAPER_B=0 SMD_BWD_TAPER_B=0
run SMD_FWD_RH=12 SMD_FWD_TAPER_B=0 SMD_BWD_TAPER_B=0
done
done
