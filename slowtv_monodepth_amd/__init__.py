"""MI355X-native view-synthesis loss hot path of self-supervised monocular depth training (gfx950 HIP kernels behind
the registry/cfg operator surface of jspenmar/slowtv_monodepth).  See DESIGN.md and INTEGRATION.md.

Importing the package loads `libsmd_hotpath.so` (built in-tree by `__graft_entry__.build()`); there is no fallback."""
from . import miopen_tuning
miopen_tuning.install()   # before any convolution runs in this process
from . import _lib, functional, geometry, handlers, io, losses, networks, ops, parsers, registry, regularizers  # noqa: F401
from .registry import DEC_REG, LOSS_REG, NET_REG, register  # noqa: F401

__version__ = '0.1.0'
