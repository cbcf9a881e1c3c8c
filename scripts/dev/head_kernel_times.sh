cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/hp && rocprofv3 --kernel-trace --output-format csv -d /tmp/hp -o t -- python $GRAFT_REPO_ROOT/scripts/dev/decoder_conv_times.py > /dev/null 2>&1; f=$(find /tmp/hp -name "*kernel_trace.csv" | head -1); python - "$f" <<PY
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    if "k_head" in r["Kernel_Name"] or "k_thin" in r["Kernel_Name"]:
        k = (r["Kernel_Name"][5:30], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])
        agg.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))/1e3)
for k, v in agg.items(): print(k, len(v), "median", sorted(v)[len(v)//2])
PY
