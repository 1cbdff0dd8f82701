"""ctypes binding of libstlpose_hip.so (see include/stlpose_hip.h).

The product path has no CPU fallback: if the HIP library is missing this module raises at
import-of-symbols time (``lib()``), loudly.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("STLPOSE_HIP_LIB") or os.path.join(_HERE, "libstlpose_hip.so")   # override: A/B of two builds

F32, BF16, F16 = 0, 1, 2


def dt2(grad_dtype: int, fwd_dtype: int) -> int:
    """STL_DT2: two element types in one int (low byte: gradient-side tensors, bits 8-15: forward-side tensors if different)."""
    return grad_dtype | ((fwd_dtype if fwd_dtype != grad_dtype else 0) << 8)


MIXED = dt2(BF16, F16)   # the mixed 16-bit mode: forward tensors f16, gradients bf16
NSHARD = 2
WGRAD_GROUP_MAX = 8
SRC_PLAIN, SRC_BN, SRC_BNBWD, SRC_BNADD = 0, 1, 2, 3
vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float


class Src(C.Structure):
    _fields_ = [("x", vp), ("y", vp), ("mode", i32), ("relu", i32), ("stats", vp), ("rstats", vp),
                ("gamma", vp), ("beta", vp), ("rmean", vp), ("rvar", vp), ("inv_count", f32), ("eps", f32)]


class Conv(C.Structure):
    _fields_ = [("dtype", i32), ("B", i32), ("Hi", i32), ("Wi", i32), ("Ci", i32), ("Ho", i32), ("Wo", i32),
                ("Co", i32), ("ks", i32), ("stride", i32), ("stuff", i32), ("TH", i32), ("TW", i32), ("shape", i32),
                ("src", Src), ("w", vp), ("out", vp), ("bias", vp), ("out_relu", i32), ("out_stats", vp),
                ("addend", vp), ("mask_y", vp), ("mask_bn", Src), ("red", vp), ("mask_z", vp),
                ("src_out", vp), ("ydtype", i32), ("pad_", i32)]


class Wgrad(C.Structure):
    _fields_ = [("dtype", i32), ("B", i32), ("Hi", i32), ("Wi", i32), ("Ci", i32), ("Ho", i32), ("Wo", i32),
                ("Co", i32), ("ks", i32), ("stride", i32), ("TH", i32), ("TW", i32), ("nsplit", i32),
                ("h", Src), ("g", Src), ("partial", vp), ("ydtype", i32)]


class WgradGroup(C.Structure):
    _fields_ = [("n", i32), ("pad_", i32), ("p", C.POINTER(Wgrad) * 8)]


class Term(C.Structure):
    _fields_ = [("src", Src), ("shift", i32)]


class Fuse(C.Structure):
    _fields_ = [("dtype", i32), ("B", i32), ("H", i32), ("W", i32), ("C", i32), ("nterms", i32), ("relu", i32),
                ("t", Term * 4), ("out", vp)]


class FuseBwd(C.Structure):
    _fields_ = [("dtype", i32), ("B", i32), ("H", i32), ("W", i32), ("C", i32), ("ngrads", i32), ("relu", i32),
                ("nbn", i32), ("dz", vp * 4), ("z", vp), ("du", vp), ("bn", Src * 4), ("rstats", vp * 4), ("ydtype", i32), ("pad_", i32)]


class UpBwd(C.Structure):
    _fields_ = [("dtype", i32), ("B", i32), ("H", i32), ("W", i32), ("C", i32), ("shift", i32),
                ("du", vp), ("dt", vp), ("bn", Src), ("rstats", vp), ("ydtype", i32), ("pad_", i32)]


class WPrep(C.Structure):
    _fields_ = [("src_off", i64), ("fwd_off", i64), ("bwd_off", i64), ("Co", i32), ("Ci", i32), ("ks", i32),
                ("Cip", i32), ("patch", i32), ("blk0", i32)]


class Slab(C.Structure):
    _fields_ = [("part_off", i64), ("grad_off", i64), ("nsplit", i32), ("Co", i32), ("Ci", i32), ("ks", i32),
                ("Cip", i32), ("patch", i32), ("blk0", i32), ("pad", i32)]


class BNRec(C.Structure):
    _fields_ = [("stats_off", i64), ("param_off", i64), ("buf_off", i64), ("C", i32), ("inv_count", f32)]


class Patch(C.Structure):
    _fields_ = [("dtype", i32), ("B", i32), ("H", i32), ("W", i32), ("stride", i32), ("pad_", i32), ("img", vp), ("out", vp),
                ("mean3", vp), ("std3", vp)]


class Head(C.Structure):
    _fields_ = [("dtype", i32), ("B", i32), ("H", i32), ("W", i32), ("Ci", i32), ("J", i32), ("x", vp), ("w", vp),
                ("bias", vp), ("out", vp)]


class HeadBwd(C.Structure):
    _fields_ = [("dtype", i32), ("B", i32), ("H", i32), ("W", i32), ("Ci", i32), ("J", i32), ("nblk", i32), ("pad_", i32),
                ("x", vp), ("w", vp), ("dout", vp), ("dx", vp), ("partial", vp)]


class Op(C.Structure):
    _fields_ = [("kind", i32), ("stream", i32), ("desc", vp), ("nwait", i32), ("wait", i32 * 8), ("record", i32)]


class ReduceRange(C.Structure):
    _fields_ = [("partials", vp), ("grads", vp), ("tab", vp), ("n", i32), ("blk_base", i32), ("nblocks", i32), ("pad_", i32)]


class BNRange(C.Structure):
    _fields_ = [("rstats", vp), ("grads", vp), ("tab", vp), ("n", i32), ("pad_", i32)]


class OptimSlice(C.Structure):   # stl_optim_slice
    _fields_ = [("kind", i32), ("pad_", i32), ("p", vp), ("g", vp), ("m", vp), ("v", vp), ("n", C.c_int64), ("hyper", vp), ("step", vp)]


class WPrepRange(C.Structure):   # stl_wprep_range
    _fields_ = [("dtype", i32), ("n", i32), ("blk_base", i32), ("nblocks", i32), ("master", vp), ("wk", vp), ("tab", vp)]


OP_KIND = {"stl_conv_forward": 0, "stl_conv_wgrad": 1, "stl_fuse_forward": 2, "stl_fuse_backward": 3,
           "stl_upsample_backward": 4, "stl_patch3x3": 5, "stl_head_forward": 6, "stl_head_backward": 7,
           "stl_reduce_slabs_range": 8, "stl_bn_grads_range": 9, "stl_conv_wgrad_group": 10,
           "stl_optim_slice": 11, "stl_wprep_range": 12}   # the last two exist as program ops only

# name -> argtypes (restype is always int unless noted); every symbol include/stlpose_hip.h declares
SIGNATURES = {
    "stl_conv_forward": [C.POINTER(Conv), vp],
    "stl_conv_plan": [C.POINTER(Conv)],
    "stl_conv_bnadd_ok": [C.POINTER(Conv)],
    "stl_conv_wgrad": [C.POINTER(Wgrad), vp],
    "stl_wgrad_chunk": [C.POINTER(Wgrad)],
    "stl_conv_wgrad_group": [C.POINTER(WgradGroup), vp],
    "stl_fuse_forward": [C.POINTER(Fuse), vp],
    "stl_fuse_backward": [C.POINTER(FuseBwd), vp],
    "stl_upsample_backward": [C.POINTER(UpBwd), vp],
    "stl_patch3x3": [i32, vp, vp, i32, i32, i32, i32, vp, vp, vp],
    "stl_head_forward": [i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "stl_head_backward": [i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "stl_mse_loss": [vp, vp, vp, vp, vp, i32, vp, i32, i32, i32, f32, vp],
    "stl_gaussian_targets": [vp, vp, vp, vp, i32, i32, i32, i32, f32, f32, f32, vp],
    "stl_heatmap_argmax": [vp, vp, vp, vp, i32, i32, i32, vp],
    "stl_flip_merge": [vp, vp, vp, vp, i32, i32, i32, i32, vp],
    "stl_final_preds": [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp],
    "stl_weight_prep": [i32, vp, vp, vp, i32, i32, vp],
    "stl_weight_prep_range": [i32, vp, vp, vp, i32, i32, i32, vp],
    "stl_optim_begin_step": [vp, vp, vp],
    "stl_adam_slice": [vp, vp, vp, vp, i64, vp, vp, vp],
    "stl_sgd_slice": [vp, vp, vp, i64, vp, vp, vp],
    "stl_reduce_slabs": [vp, vp, vp, i32, i32, vp],
    "stl_reduce_slabs_range": [C.POINTER(ReduceRange), vp],
    "stl_bn_running_update": [vp, vp, vp, vp, i32, f32, vp, vp],
    "stl_bn_param_grads": [vp, vp, vp, i32, vp],
    "stl_adam_step": [vp, vp, vp, vp, i64, vp, vp, vp, vp],
    "stl_sgd_step": [vp, vp, vp, i64, vp, vp, vp, vp],
    "stl_affine_crop": [vp, vp, vp, vp, vp, vp, i32, i32, i32, vp, vp, vp],
    "stl_maxpool2x2": [i32, vp, vp, i32, i32, i32, i32, vp],
    "stl_l1_partial": [i32, vp, vp, i64, vp, i32, vp],
    "stl_l2_partial": [i32, vp, vp, i64, vp, i32, vp],
    "stl_bilinear_nchw": [vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "stl_sum_partials": [vp, i32, C.c_double, vp, i32, vp],
    "stl_nchw_to_nhwc": [i32, vp, vp, i32, i32, i32, i32, vp],
    "stl_nhwc_to_nchw": [i32, vp, vp, i32, i32, i32, i32, vp],
    "stl_program_create": [C.POINTER(Op), i32, i32, C.POINTER(vp)],
    "stl_program_run": [vp, C.POINTER(vp)],
    "stl_program_run_range": [vp, C.POINTER(vp), i32, i32],
    "stl_program_destroy": [vp],
    "stl_program_wait_op": [vp, i32, vp],
    "stl_program_graph_build": [vp],
    "stl_program_graph_launch": [vp, vp],
    "stl_selftest_mfma": [vp, vp],
    "stl_version": [],
}

# include/stlpose_hip_debug.h: exported by the stamped diagnostic build only (STLPOSE_HIP_LIB=.../libstlpose_hip_stamps.so)
DEBUG_SIGNATURES = {"stl_debug_conv_stamps": [vp], "stl_debug_conv_stamps2": [vp], "stl_debug_wgrad_stamps": [vp],
                    "stl_debug_wgrad_stamps2": [vp]}

STRING_FUNCS = ("stl_last_error", "stl_build_id", "stl_last_kernel")   # const char* f(void)

_lib = None


def lib() -> C.CDLL:
    """Load the HIP library (once).  Raises RuntimeError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -m stlpose_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback for the product path.")
        l = C.CDLL(LIB_PATH)
        for name, args in SIGNATURES.items():
            fn = getattr(l, name)  # AttributeError if the ABI and this table disagree
            fn.argtypes = args
            fn.restype = C.c_int
        for name, args in DEBUG_SIGNATURES.items():
            if hasattr(l, name):
                getattr(l, name).argtypes, getattr(l, name).restype = args, C.c_int
        for name in STRING_FUNCS:
            fn = getattr(l, name)
            fn.argtypes = []
            fn.restype = C.c_char_p
        _lib = l
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        raise RuntimeError(f"stlpose_hip {what}: {lib().stl_last_error().decode()}")


def call(name: str, *args) -> None:
    check(getattr(lib(), name)(*args), name)
