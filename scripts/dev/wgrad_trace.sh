#!/bin/bash
# kernel trace of the weight-gradient entry point alone (GPU box): wgrad_trace.sh C CO h w [b] [pieces]
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_w
timeout -k 5 100 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_w -o w -- python $GRAFT_REPO_ROOT/scripts/dev/wgrad_only.py "$@" > /tmp/prof_w.log 2>&1
f=$(find /tmp/prof_w -name "*kernel_stats.csv" | head -1)
echo "== $*"; grep -E "wgrad|finalize|thin_wgt" $f | awk -F'","' '{print $1, "calls", $2, "avg_ns", $4}' | cut -c1-200
