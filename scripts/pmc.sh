#!/bin/bash
# Collect rocprofv3 PMC counters for the hot-path kernels (one pass per counter group), on the GPU box.
# usage: scripts/pmc.sh <tag> <cmd...>      -> gpurun_out/pmc_<tag>/summary.txt
set -u
tag=$1; shift
export TMPDIR=/tmp
out=$PWD/gpurun_out/pmc_$tag
mkdir -p "$out"
if [ -n "${PMC_GROUPS:-}" ]; then IFS=';' read -ra groups <<< "$PMC_GROUPS"; else
groups=(
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM"
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_LDS GRBM_GUI_ACTIVE"
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum"
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
  "FETCH_SIZE"
  "WRITE_SIZE"
)
fi
i=0
for g in "${groups[@]}"; do
  ( cd /tmp && timeout -k 5 ${PMC_TIMEOUT:-90} rocprofv3 --pmc $g --kernel-trace --output-format csv -d "$out/raw$i" -o pmc -- "$@" > "$out/run$i.log" 2>&1 )
  i=$((i+1))
done
python "$GRAFT_REPO_ROOT/scripts/summarize_pmc.py" "$out" > "$out/summary.txt" 2>&1
rm -rf "$out"/raw*
cat "$out/summary.txt"
