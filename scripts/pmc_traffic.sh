#!/bin/bash
# HBM traffic of the two fused kernels measured on bench.py's OWN inputs (GPU box): separate rocprofv3 --pmc passes for
# FETCH_SIZE, WRITE_SIZE and the L2 hit/miss counters over `bench.py --workload <wl>`, then profiles/traffic.json is regenerated.
# usage: scripts/pmc_traffic.sh <tag> [workload ...]        (default workload: cfg2)
set -u
tag=$1; shift
wls=("$@"); [ ${#wls[@]} -eq 0 ] && wls=(cfg2)
root=${GRAFT_REPO_ROOT:-$PWD}
for wl in "${wls[@]}"; do
  PMC_TIMEOUT=${PMC_TIMEOUT:-240} PMC_GROUPS="FETCH_SIZE;WRITE_SIZE;TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
    bash "$root/scripts/pmc.sh" "${tag}_bench_$wl" python "$root/bench.py" --workload "$wl" --steps 3 --warmup 2 --no-cpu-baseline > /dev/null
done
python "$root/scripts/make_traffic_json.py" "$tag" "${wls[@]}" > "$root/gpurun_out/traffic_$tag.json"
cat "$root/gpurun_out/traffic_$tag.json"
