#!/bin/bash
# rocprofv3 kernel-trace of scripts/dev/microbench.py (GPU box), CSV output, summarised per kernel.  usage: prof_micro_csv.sh <tag> [cfg] (env passes through)
tag=$1; cfg=${2:-cfg2}
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/mprof_$tag
mkdir -p "$out"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/raw" -o trace -- python "$GRAFT_REPO_ROOT/scripts/dev/microbench.py" $cfg 20 > "$out/run.log" 2>&1
cd "$GRAFT_REPO_ROOT"
find "$out/raw" -name '*kernel_stats.csv' -exec cp {} "$out/kernel_stats.csv" \;
rm -rf "$out/raw"
python - "$out/kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if 'smd::' in r['Name']: print(f"{r['Name'][:90]:90s} n={r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:8.2f} us  min {float(r['MinNs'])/1e3:8.2f}")
PY
