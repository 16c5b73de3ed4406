"""ctypes binding of include/hubbardtn_hip.h.

The product path has exactly one implementation of the device primitives: libhubbardtn_hip.so.
`load_library()` raises if the shared object is missing -- there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "csrc", "libhubbardtn_hip.so")

HTN_MAX_BUFS = 8
HTN_TILE = 32
OP_N, OP_T, OP_C = 0, 1, 2
SEG_GEMM, SEG_COPY = 0, 1

# numpy mirrors of the C structs (sizes asserted against the header's comments)
TILE_DT = np.dtype([("c_off", "<i8"), ("buf_c", "<i4"), ("ldc", "<i4"), ("m", "<i4"), ("n", "<i4"),
                    ("row0", "<i4"), ("col0", "<i4"), ("seg_begin", "<i4"), ("seg_count", "<i4"),
                    ("pad0", "<i4"), ("pad1", "<i4")], align=False)
SEG_DT = np.dtype([("a_off", "<i8"), ("b_off", "<i8"), ("buf_a", "<i4"), ("buf_b", "<i4"),
                   ("lda", "<i4"), ("ldb", "<i4"), ("k", "<i4"), ("op_a", "<i4"), ("op_b", "<i4"),
                   ("type", "<i4"), ("alpha_re", "<f8"), ("alpha_im", "<f8")], align=False)
SVD_DT = np.dtype([("g_off", "<i8"), ("v_off", "<i8"), ("s_off", "<i8"), ("m", "<i4"), ("n", "<i4"),
                   ("flags", "<i4"), ("pad", "<i4")], align=False)
SVD_ACCUMULATE = 1
SVD_QRCP = 2
COPY_DT = np.dtype([("dst_off", "<i8"), ("src_off", "<i8"), ("idx_off", "<i8"), ("scl_off", "<i8"),
                    ("rows", "<i4"), ("cols", "<i4"), ("ldd", "<i4"), ("lds", "<i4"), ("op", "<i4"),
                    ("gather_dim", "<i4"), ("scale_dim", "<i4"), ("inv_norm", "<i4")], align=False)
assert TILE_DT.itemsize == 48 and SEG_DT.itemsize == 64 and SVD_DT.itemsize == 40 and COPY_DT.itemsize == 64

EXPORTS = ["htn_last_error", "htn_abi_version", "htn_device_init", "htn_grouped_gemm_z",
           "htn_dots_scratch_elems", "htn_dots_z", "htn_axpys_z", "htn_scale_inv_sqrt_z",
           "htn_jacobi_svd_z", "htn_jacobi_set_split", "htn_jacobi_set_rank_cut", "htn_batched_copy_z", "htn_lanczos_scratch_elems", "htn_lanczos_z"]


class GemmLaunch(C.Structure):
    """htn_gemm_launch"""
    _fields_ = [("bufs", C.c_void_p * HTN_MAX_BUFS), ("tiles", C.c_void_p), ("segs", C.c_void_p),
                ("n_tiles", C.c_int32), ("pad", C.c_int32)]


EXCHANGE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int64, C.c_void_p)

_lib = None


class HtnError(RuntimeError):
    pass


def load_library(path: str | None = None):
    """dlopen libhubbardtn_hip.so and declare prototypes.  Fails loudly when it is absent."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    # torch ships its own libamdhip64; it must be in the process BEFORE our library is dlopen'ed so
    # that both bind to ONE HIP runtime (two runtimes in one process cannot both see the device)
    import torch  # noqa: F401
    if not os.path.exists(p):
        raise HtnError(f"{p} not found: build it with `python -m hubbardtn_amd.build` "
                       "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    lib = C.CDLL(p)
    vp, i32, i64, f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_double
    lib.htn_last_error.restype = C.c_char_p
    lib.htn_abi_version.restype = C.c_int
    lib.htn_device_init.argtypes = [C.c_int, C.c_char_p, C.POINTER(C.c_int)]
    lib.htn_grouped_gemm_z.argtypes = [C.POINTER(vp), vp, i32, vp, vp]
    lib.htn_dots_scratch_elems.argtypes = [i32]
    lib.htn_dots_scratch_elems.restype = i64
    lib.htn_dots_z.argtypes = [vp, i64, i32, vp, i64, vp, vp, vp]
    lib.htn_axpys_z.argtypes = [vp, vp, i64, i32, vp, f64, i64, vp]
    lib.htn_scale_inv_sqrt_z.argtypes = [vp, vp, vp, i64, vp]
    lib.htn_jacobi_svd_z.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, f64, vp, vp]
    lib.htn_jacobi_set_split.argtypes = [i32]
    lib.htn_jacobi_set_split.restype = i32
    lib.htn_jacobi_set_rank_cut.argtypes = [f64]
    lib.htn_jacobi_set_rank_cut.restype = f64
    lib.htn_batched_copy_z.argtypes = [vp, vp, vp, vp, vp, i32, f64, vp]
    lib.htn_lanczos_scratch_elems.argtypes = [i32]
    lib.htn_lanczos_scratch_elems.restype = i64
    lib.htn_lanczos_z.argtypes = [C.POINTER(GemmLaunch), i32, i32, i32, vp, i64, i32, f64, i32, vp, i32,
                                  EXCHANGE_FN, vp, C.POINTER(f64), C.POINTER(i32), C.POINTER(f64), C.POINTER(f64), vp]
    for name in EXPORTS:
        getattr(lib, name)          # raises AttributeError if a declared symbol is missing
    if lib.htn_abi_version() != 1:
        raise HtnError("ABI version mismatch")
    if path is None:
        _lib = lib
    return lib


def check(lib, rc: int, what: str):
    if rc != 0:
        raise HtnError(f"{what}: {lib.htn_last_error().decode()}")
