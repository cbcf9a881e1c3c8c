#!/bin/bash
# Ablation of the four-support backward at cfg 5 (GPU box): SMD_ABLATE_BWD bit 0 no tap gathers, bit 1 no target-side row loads (py / ta / tb / sel / depth ahead),
# bit 2 no g_depth traffic, bit 3 no LDS history.  Times are NOT numerically meaningful results; they bound what removing a stream can buy.
# The variant libraries are built on the CPU box first (hipcc on the GPU box was not reliable inside a gpurun call):
#   cd slowtv_monodepth_amd/csrc; for a in 2 1 3 8; do rm -f smd_recon_bwd.o; make -s EXTRA=-DSMD_ABLATE_BWD=$a; cp ../libsmd_hotpath.so ../../scripts/dev/_abl/libsmd_abl$a.so; done; rm -f smd_recon_bwd.o; make -s
cd "$GRAFT_REPO_ROOT"
for abl in 0 2 1 3 8; do
  lib=$GRAFT_REPO_ROOT/slowtv_monodepth_amd/libsmd_hotpath.so; [ $abl != 0 ] && lib=$GRAFT_REPO_ROOT/scripts/dev/_abl/libsmd_abl$abl.so
  for kn in "" "bwd_wps=4"; do
    echo -n "SMD_ABLATE_BWD=$abl ${kn:-default}: "
    SMD_HOTPATH_LIB=$lib MB_KNOBS=$kn MB_PATH=node timeout 200 python scripts/dev/microbench.py cfg5 20 2>&1 | tail -1 | sed 's/.*| bwd med/bwd med/' | cut -c1-60
  done
done
