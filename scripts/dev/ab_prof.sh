#!/bin/bash
# A/B kernel-level timing on ONE box: rocprofv3 kernel trace of the loss-path microbenchmark in two source trees.
# usage: scripts/dev/ab_prof.sh <tag> <treeA> <treeB> [cfg] [extra env as VAR=val ...]
set -u
tag=$1; A=$2; B=$3; cfg=${4:-cfg2}; shift 4 || true
export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/ab_$tag
mkdir -p "$out"
for t in A B; do
  tree=$([ $t = A ] && echo "$A" || echo "$B")
  for rough in 0 1; do
    ( cd /tmp && env "$@" MB_ROUGH=$rough timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/raw_${t}_$rough" -o t -- python "$root/$tree/scripts/dev/microbench.py" "$cfg" 30 > "$out/${t}_$rough.log" 2>&1 )
    f=$(find "$out/raw_${t}_$rough" -name '*kernel_stats.csv' | head -1)
    echo "== tree $t ($tree) rough=$rough"; grep -v amdgpu.ids "$out/${t}_$rough.log" | tail -1
    python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r['Name']
    if 'smd' in n or 'k_' in n:
        print(f"  {float(r['AverageNs'])/1e3:9.2f} us x{r['Calls']:>4}  {n[:110]}")
PY
    rm -rf "$out/raw_${t}_$rough"
  done
done
