#!/bin/bash
# rocprofv3 kernel durations of the two fused kernels for a list of environment settings (one line per setting).
# usage: scripts/dev/sweep_micro.sh <cfg> "VAR=val VAR2=val" "VAR=val" ...
cfg=$1; shift
export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-$PWD}
for envs in "$@"; do
  for rough in 0 1; do
    d=$(mktemp -d)
    ( cd /tmp && env $envs MB_ROUGH=$rough timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -o t -- python "$root/scripts/dev/microbench.py" "$cfg" 30 > "$d/log" 2>&1 )
    f=$(find "$d" -name '*kernel_stats.csv' | head -1)
    python - "$f" "$envs" "$rough" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
g = lambda key: next((float(r['AverageNs'])/1e3 for r in rows if key in r['Name']), float('nan'))
print(f"[{sys.argv[2]:<60}] rough={sys.argv[3]} main {g('k_recon_main'):7.2f}  bwd {g('k_recon_bwd'):7.2f}  prep {g('k_recon_prep'):6.2f}  k0adj_v {g('k_disp_to_depth_bwd_v'):6.2f}  smooth {g('k_smooth_main'):6.2f} us")
PY
    rm -rf "$d"
  done
done
