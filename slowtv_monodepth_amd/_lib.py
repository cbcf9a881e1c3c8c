"""ctypes binding of the C ABI in `include/smd_hotpath.h` (built in-tree as `slowtv_monodepth_amd/libsmd_hotpath.so`).

There is deliberately NO fallback: if the shared library is missing or a call fails, an exception is raised.
The library is plain `extern "C"` (raw device pointers + sizes + a stream handle); PyTorch is only used by the
callers in `functional.py` to own the device buffers and the stream.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import torch  # noqa: F401  MUST precede the CDLL below: the library binds to the HIP runtime torch has already loaded

__all__ = ['lib', 'call', 'set_knob', 'reset_knobs', 'knob_epoch', 'lib_path', 'FLAGS', 'REGR_FLAGS', 'SEL_MASKED', 'ptr_array', 'int_array', 'HotpathError', 'Unsupported']

_HERE = Path(__file__).resolve().parent
lib_path = Path(os.environ.get('SMD_HOTPATH_LIB', _HERE/'libsmd_hotpath.so'))

FLAGS = {'use_min': 0x1, 'use_automask': 0x2, 'loss_l1': 0x4, 'need_k_grad': 0x8, 'use_edges': 0x10, 'loss_l2': 0x20, 'packed_ready': 0x40,
         'mask_explainability': 0x80, 'mask_uncertainty': 0x100, 'use_laplacian': 0x200, 'bwd_skip_rows': 0x400, 'edges_ready': 0x800, 'bwd_no_live': 0x1000}
REGR_FLAGS = {'l1': 0x0, 'log_l1': 0x1, 'berhu': 0x2, 'invert': 0x4}
SEL_MASKED = 255
MAX_SCALES = 8
MAX_SUPPORTS = 8

_vp, _i, _f, _sz, _u64 = C.c_void_p, C.c_int, C.c_float, C.c_size_t, C.c_uint64

# name -> (restype, argtypes); kept in the order of include/smd_hotpath.h
PROTOTYPES = {
    'smd_last_error': (C.c_char_p, []),
    'smd_abi_version': (_i, []),
    'smd_last_kernel_variant': (C.c_char_p, [_i]),
    'smd_set_knob': (_i, [C.c_char_p, _i]),
    'smd_reset_knobs': (None, []),
    'smd_disp_to_depth_fwd': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _f, _f, _vp, _vp, _vp]),
    'smd_disp_to_depth_workspace_bytes': (_sz, [_vp, _vp, _i, _i, _i, _i]),
    'smd_disp_to_depth_bwd': (_i, [_vp, _vp, _i, _i, _i, _i, _f, _f, _vp, _vp, _vp, _vp, _sz, _vp]),
    'smd_image_recon_workspace_bytes': (_sz, [_i, _i, _i, _i, _i]),
    'smd_packed_supports_bytes': (_sz, [_i, _i, _i, _i]),
    'smd_image_recon_prep': (_i, [_vp]*5 + [_i]*6 + [_vp]),
    'smd_image_recon_supports_per_pass': (_i, []),
    'smd_image_recon_fwd': (_i, [_vp]*7 + [_u64] + [_vp]*6 + [_sz] + [_i]*6 + [_vp]),
    'smd_image_recon_bwd': (_i, [_vp]*13 + [_sz] + [_i]*6 + [_vp]),
    'smd_image_recon_disp_workspace_bytes': (_sz, [_vp, _vp, _i, _i, _i, _i, _i]),
    'smd_image_recon_disp_fwd': (_i, [_vp, _vp, _vp, _i, _f, _f] + [_vp]*6 + [_u64] + [_vp]*7 + [_sz] + [_i]*5 + [_vp]),
    'smd_image_recon_disp_bwd': (_i, [_vp, _vp, _i, _f, _f] + [_vp]*13 + [_sz] + [_i]*5 + [_vp]),
    'smd_loss_path_workspace_bytes': (_sz, [_vp, _vp, _i, _i, _i, _i, _i]),
    'smd_loss_path_fwd': (_i, [_vp]*4 + [_i, _f, _f] + [_vp]*5 + [_u64] + [_vp]*7 + [_sz] + [_i]*5 + [_f, _f, _vp]),
    'smd_loss_path_bwd': (_i, [_vp]*4 + [_i, _f, _f] + [_vp]*9 + [_f, _f] + [_vp]*14 + [_sz] + [_i]*5 + [_vp]),
    'smd_disp_smooth_workspace_bytes': (_sz, [_vp, _vp, _i, _i]),
    'smd_disp_smooth_edge_weight_bytes': (_sz, [_vp, _vp, _i, _i]),
    'smd_gaussian_blur3x3': (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    'smd_disp_smooth_prep': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    'smd_disp_smooth_fwd': (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    'smd_disp_smooth_bwd': (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    'smd_view_synth_workspace_bytes': (_sz, [_i, _i, _i]),
    'smd_view_synth_fwd': (_i, [_vp]*8 + [_i]*4 + [_vp]),
    'smd_view_synth_bwd': (_i, [_vp]*13 + [_sz] + [_i]*4 + [_vp]),
    'smd_photo_error_workspace_bytes': (_sz, [_i, _i, _i, _i]),
    'smd_photo_error_fwd': (_i, [_vp]*3 + [_i]*5 + [_f, _vp]),
    'smd_photo_error_bwd': (_i, [_vp]*5 + [_sz] + [_i]*5 + [_f, _vp]),
    'smd_regression_workspace_bytes': (_sz, [_sz]),
    'smd_regression_fwd': (_i, [_vp]*3 + [_sz, _i] + [_vp]*4 + [_sz, _vp]),
    'smd_regression_bwd': (_i, [_vp]*3 + [_sz, _i] + [_vp]*5 + [_sz, _vp]),
    'smd_recon_reduce_workspace_bytes': (_sz, [_i, _i, _i]),
    'smd_recon_reduce_fwd': (_i, [_vp]*4 + [_u64] + [_vp]*4 + [_sz] + [_i]*5 + [_vp]),
    'smd_recon_reduce_bwd': (_i, [_vp]*7 + [_i]*5 + [_vp]),
    'smd_decoder_glue_workspace_bytes': (_sz, [_i]*4),
    'smd_conv3x3_thin_workspace_bytes': (_sz, [_i]*4),
    'smd_conv3x3_thin_fwd': (_i, [_vp]*3 + [_i]*4 + [_vp]),
    'smd_conv3x3_thin_bwd': (_i, [_vp]*6 + [_sz] + [_i]*4 + [_vp]),
    'smd_conv3x3_mfma_packed_bytes': (_sz, [_i]*3),
    'smd_conv3x3_mfma_workspace_bytes': (_sz, [_i]*5),
    'smd_conv3x3_mfma_pack': (_i, [_vp]*3 + [_i]*3 + [_vp]),
    'smd_conv3x3_mfma_fwd': (_i, [_vp]*4 + [_sz] + [_i]*6 + [_vp]),
    'smd_conv3x3_mfma_bwd_data': (_i, [_vp]*4 + [_sz] + [_i]*6 + [_vp]),
    'smd_conv3x3_mfma_bwd_weight': (_i, [_vp]*4 + [_sz] + [_i]*6 + [_vp]),
    'smd_conv3x3_head_workspace_bytes': (_sz, [_i]*4),
    'smd_conv3x3_head_fwd': (_i, [_vp]*4 + [_i]*5 + [_vp]),
    'smd_conv3x3_head_bwd': (_i, [_vp]*8 + [_sz] + [_i]*5 + [_vp]),
    'smd_elu_pad_fwd': (_i, [_vp]*3 + [_i]*6 + [_vp]),
    'smd_elu_pad_bwd': (_i, [_vp]*6 + [_sz] + [_i]*6 + [_vp]),
    'smd_elu_up_cat_pad_fwd': (_i, [_vp]*4 + [_i]*6 + [_vp]),
    'smd_elu_up_cat_pad_bwd': (_i, [_vp]*7 + [_sz] + [_i]*6 + [_vp]),
    'smd_bn_workspace_bytes': (_sz, [_i, _i, _i]),
    'smd_bn_fwd': (_i, [_vp]*6 + [_f, _f, _i] + [_vp]*4 + [_sz] + [_i]*3 + [_vp]),
    'smd_bn_bwd': (_i, [_vp]*6 + [_i] + [_vp]*5 + [_sz] + [_i]*3 + [_vp]),
    'smd_maxpool3x3s2_fwd': (_i, [_vp]*3 + [_i]*4 + [_vp]),
    'smd_maxpool3x3s2_bwd': (_i, [_vp]*3 + [_i]*4 + [_vp]),
    'smd_dwconv7x7_workspace_bytes': (_sz, [_i]*3),
    'smd_dwconv7x7_fwd': (_i, [_vp]*4 + [_i]*5 + [_vp]),
    'smd_dwconv7x7_wrw': (_i, [_vp]*5 + [_sz] + [_i]*4 + [_vp]),
    'smd_layernorm_cf_workspace_bytes': (_sz, [_i]*3),
    'smd_layernorm_cf_fwd': (_i, [_vp]*4 + [_i] + [_vp]*2 + [_i]*3 + [_f, _vp]),
    'smd_layernorm_cf_bwd': (_i, [_vp]*2 + [_i] + [_vp]*7 + [_sz] + [_i]*3 + [_vp]),
    'smd_pose_fwd': (_i, [_vp, _vp, _vp, _i, _vp, _vp]),
    'smd_pose_bwd': (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    'smd_intrinsics_fwd': (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    'smd_intrinsics_bwd': (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    'smd_crop_resize': (_i, [_vp, _vp, _vp] + [_i]*7 + [_vp, _vp, _i, _vp]),
    'smd_profile_enable': (_i, [_i, _i]),
    'smd_profile_collect': (_i, [_i, _vp, _i, _vp]),
    'smd_debug_stream_copy': (_i, [_vp, _vp, _sz, _i, _vp]),
    'smd_debug_lane_shift': (_i, [_vp, _vp, _vp]),
}


class HotpathError(RuntimeError):
    """A call into libsmd_hotpath.so returned a negative status."""


class Unsupported(HotpathError):
    """SMD_E_UNSUPPORTED: the entry point does not serve these arguments in this build (nothing was launched); the caller takes the general operators."""


def _load() -> C.CDLL:
    if not lib_path.is_file():
        raise ImportError(
            f'{lib_path} not found: the HIP hot path is not built. Run `python -c "import __graft_entry__ as g; g.build()"` '
            f'(or `make -C slowtv_monodepth_amd/csrc`). There is no CPU/PyTorch fallback for this path.')
    handle = C.CDLL(str(lib_path))
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(handle, name)  # AttributeError if the symbol is not exported
        fn.restype, fn.argtypes = res, args
    if handle.smd_abi_version() != 8: raise ImportError(f'ABI version mismatch in {lib_path}')
    return handle


lib = _load()


def call(name: str, *args):
    """Invoke an int-returning entry point; map SMD_E_INVALID to ValueError (the reference's exception type for bad
    arguments) and everything else to HotpathError."""
    rc = getattr(lib, name)(*args)
    if rc != 0:
        msg = lib.smd_last_error().decode()
        if rc == -1: raise ValueError(f'{name}: {msg}')
        if rc == -4: raise Unsupported(f'{name}: {msg}')
        raise HotpathError(f'{name} failed ({rc}): {msg}')


def set_knob(name: str, value: int) -> bool:
    """Pin a launch-shape knob of the library (`smd_set_knob`: partitions / code paths that must give identical results; the parity tests
    compare both sides).  -> False if this build does not have the knob (experiments-only), ValueError for an unknown name."""
    global knob_epoch
    rc = lib.smd_set_knob(name.encode(), int(value))
    if rc == -1: raise ValueError(lib.smd_last_error().decode())
    knob_epoch += 1
    return rc == 0


def reset_knobs() -> None:
    global knob_epoch
    lib.smd_reset_knobs()
    knob_epoch += 1


# Every knob change in this process passes through the two functions above.  A forward call records the epoch it ran under; a backward that finds another
# one passes FLAGS['bwd_no_live']: the forward's liveness table was written under the forward's strip partition, which the backward re-derives from the
# knobs in force when IT runs (with other knobs live waves would be read as dead and their gradients silently become zeros).
knob_epoch = 0


def ptr_array(ptrs):
    return (C.c_void_p*len(ptrs))(*ptrs)


def int_array(vals):
    return (C.c_int*len(vals))(*[int(v) for v in vals])
