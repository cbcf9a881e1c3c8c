"""MI355X-native view-synthesis loss hot path of self-supervised monocular depth training (gfx950 HIP kernels behind
the registry/cfg operator surface of jspenmar/slowtv_monodepth).  See DESIGN.md."""
