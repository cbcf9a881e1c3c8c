#!/bin/bash
# Waves per strip of the fused backward (knob bwd_wps) against launch size and liveness, one box (GPU box) -> gpurun_out/bwd_wps_sweep.txt
out=gpurun_out/bwd_wps_sweep.txt; : > $out
live() { echo "== live masks (scripts/dev/cfg4_live_masks.py): n=$1 b=$2 h=$3 bwd_wps=$4" >> $out; MB_KNOBS=bwd_wps=$4 python scripts/dev/cfg4_live_masks.py $1 $2 $3 2>&1 | grep "live masks:\|row loop" >> $out; }
train() { echo "== masks of a 12-step training run (scripts/dev/bwd_live_ab.py): $1 bwd_wps=$2" >> $out; MB_KNOBS=bwd_wps=$2 python scripts/dev/bwd_live_ab.py $1 12 2>&1 | grep "after 12\|row loop" >> $out; }
for w in 4 2 1; do train cfg5 $w; done
for w in 2 1; do train cfg4 $w; train cfg2 $w; done
for w in 4 1; do live 4 12 384 $w; live 4 6 384 $w; live 4 12 192 $w; done
for w in 3 1; do live 3 12 384 $w; done
for w in 2 1; do live 2 12 384 $w; live 2 6 384 $w; live 2 12 192 $w; live 2 24 192 $w; done
cat $out
