#!/bin/bash
# Median rocprofv3 duration of every launch shape (kernel, grid) whose name contains $1, over `python bench.py --steps 6 --warmup 3` (GPU box).
# usage: scripts/dev/kernel_times_by_grid.sh k_elu [bench args...]
pat=$1; shift
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt && rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline "$@" > /dev/null 2>&1
f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python - "$f" "$pat" <<PY
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    if sys.argv[2] in r["Kernel_Name"]:
        k = (r["Kernel_Name"][:70], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])
        agg.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))/1e3)
tot = 0
for k, v in sorted(agg.items(), key=lambda kv: -sorted(kv[1])[len(kv[1])//2]*len(kv[1])):
    med = sorted(v)[len(v)//2]; per_step = med*len(v)/9
    tot += per_step
    print(f"{k[0]:70s} grid {k[1]:>9s} x {k[2]:>4s} x {k[3]:>4s}  n={len(v):3d}  median {med:8.2f} us  ({per_step:7.1f} us per step)")
print(f"total {tot:.1f} us per step")
PY
