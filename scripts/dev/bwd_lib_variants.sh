#!/bin/bash
# A/B of the fused backward's row-loop variants on ONE box (GPU box): one library per -D combination, selected at run time through
# SMD_HOTPATH_LIB (slowtv_monodepth_amd/_lib.py; a diagnosis switch).  usage: scripts/dev/bwd_lib_variants.sh "tag:-DFLAGS" ...
cd "$GRAFT_REPO_ROOT/slowtv_monodepth_amd/csrc"
mkdir -p ../variants
for spec in "$@"; do
  tag=${spec%%:*}; defs=${spec#*:}
  rm -f smd_recon_bwd.o; make -s EXPERIMENTS=1 EXTRA="$defs" >/dev/null 2>&1 && cp ../libsmd_hotpath.so ../variants/libsmd_$tag.so
done
rm -f smd_recon_bwd.o; make -s >/dev/null 2>&1
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
  for spec in "$@"; do
    tag=${spec%%:*}
    for skip in ${SKIPS:-0}; do
      echo -n "[$tag skip=$skip] "
      SMD_HOTPATH_LIB=slowtv_monodepth_amd/variants/libsmd_$tag.so SMD_BWD_SKIP=$skip timeout 200 python scripts/dev/microbench.py ${CFG:-cfg2} 20 2>&1 | tail -1 | sed 's/.*| bwd med/bwd med/' | cut -c1-70
    done
  done
done
