#!/bin/bash
# kernel-trace of the loss-path microbenchmark (GPU box). usage: scripts/prof_micro.sh <tag> [cfg]
set -u
tag=${1:-m}; cfg=${2:-cfg2}
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_$tag
mkdir -p "$out"
( cd /tmp && timeout -k 5 120 rocprofv3 --kernel-trace --output-format csv -d "$out/raw" -o trace -- python "$GRAFT_REPO_ROOT/scripts/dev/microbench.py" $cfg 10 > "$out/run.log" 2>&1 )
find "$out/raw" -name '*kernel_trace.csv' -exec sh -c 'python scripts/summarize_trace.py "$1" > "$2"' _ {} "$out/trace_summary.txt" \;
rm -rf "$out/raw"
grep -v "^W2026\|^E2026" "$out/run.log" | tail -2
cut -c1-200 "$out/trace_summary.txt" | head -40
