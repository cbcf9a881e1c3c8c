#!/bin/bash
# Same-box A/B of prebuilt library variants (slowtv_monodepth_amd/variants/libsmd_<tag>.so; "cur" = the in-tree library).  (GPU box)
# usage: scripts/dev/ab_libs.sh tag [tag ...]     env: CFGS="cfg2 cfg5"  REPS=3  SKIPS="0"
cd "$GRAFT_REPO_ROOT"
for rep in $(seq 1 ${REPS:-3}); do
  for cfg in ${CFGS:-cfg2}; do
    for tag in "$@"; do
      lib=slowtv_monodepth_amd/variants/libsmd_$tag.so; [ "$tag" = cur ] && lib=slowtv_monodepth_amd/libsmd_hotpath.so
      for skip in ${SKIPS:-0}; do
        echo -n "[$tag $cfg skip=$skip] "
        SMD_HOTPATH_LIB=$lib SMD_BWD_SKIP=$skip timeout 200 python scripts/dev/microbench.py $cfg 20 2>&1 | tail -1 | sed 's/ | entry points.*//' | cut -c1-160
      done
    done
  done
done
