#!/bin/bash
# Wave traces of the forward at the launch shapes of profiles/r04_fwd_shape_sweep.txt (GPU box; needs slowtv_monodepth_amd/variants/libsmd_trace.so,
# the library built with EXTRA=-DSMD_TRACE_WAVES).
cd "$GRAFT_REPO_ROOT"
for e in "A=1" "SMD_FWD_TAPER_B=0" "SMD_FWD_RH=28 SMD_FWD_TAPER_B=9 SMD_FWD_TAPER_RH=24" "SMD_FWD_RH=32 SMD_FWD_TAPER_B=0" "SMD_FWD_RH=48 SMD_FWD_TAPER_B=0"; do
  echo "=== $e"
  env $e SMD_HOTPATH_LIB=slowtv_monodepth_amd/variants/libsmd_trace.so SMD_BWD_SKIP=0 timeout 200 python scripts/dev/wave_trace.py cfg2 2>&1 | grep -v amdgpu | grep "^waves\|^distinct\|^resident\|^time-averaged\|^start time\|mean life by start\|mean life by SIMD\|fwd med" | cut -c1-330
done
