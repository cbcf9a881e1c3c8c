#!/bin/bash
# Run on the GPU box (through gpurun): kernel-trace profile of the bench command; summary copied to gpurun_out/.
# usage: scripts/gpu_profile.sh <tag> [bench args...]
set -u
tag=${1:-r01}; shift || true
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp
timeout ${PROF_TIMEOUT:-600} rocprofv3 --kernel-trace --stats --output-format csv -d "$out/raw" -o trace -- python "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline "$@" > "$out/bench.log" 2>&1
cd "$GRAFT_REPO_ROOT"
find "$out/raw" -name '*kernel_stats.csv' -exec cp {} "$out/kernel_stats.csv" \;
find "$out/raw" -name '*kernel_trace.csv' -exec sh -c 'python scripts/summarize_trace.py "$1" > "$2"; python scripts/summarize_trace.py "$1" --steady k_recon_bwd 5 > "$3"; python scripts/summarize_trace.py "$1" --timeline k_pose_fwd k_disp_to_depth_bwd_h 4 > "$4" 2>&1' _ {} "$out/trace_summary.txt" "$out/steady_summary.txt" "$out/loss_path_timeline.txt" \;
find "$out/raw" -type f | head -20; rm -rf "$out/raw"
tail -3 "$out/bench.log"
head -50 "$out/steady_summary.txt" | cut -c1-200
