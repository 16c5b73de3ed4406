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
                    ("pad0", "<i4"), ("pad1", "<i4"), ("part", "<i4"), ("nparts", "<i4"), ("ws_slot", "<i4"),
                    ("ticket", "<i4")], align=False)
BUF_WS = 7                          # buffer-table slot of the split-K workspace
WS_TICKET_ELEMS = 4096              # complex128 elements reserved for the tickets
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
assert TILE_DT.itemsize == 64 and SEG_DT.itemsize == 64 and SVD_DT.itemsize == 40 and COPY_DT.itemsize == 64

# ---- bond-update / sweep level (ABI 2) ----
SYM_SU2_U1, SYM_U1_U1, SYM_SU2 = 0, 1, 2
MAX_SITE = 4
BACKEND_CPU, BACKEND_HIP = 0, 1
SITE_OP_DT = np.dtype([("k", "<i4"), ("dN", "<i4"), ("red", "<f8", (MAX_SITE * MAX_SITE,))], align=False)
MPO_ENTRY_DT = np.dtype([("wl", "<i4"), ("wr", "<i4"), ("op", "<i4"), ("pad", "<i4"), ("coef_re", "<f8"),
                         ("coef_im", "<f8")], align=False)
SUBBLOCK_DT = np.dtype([("lN", "<i4"), ("lj", "<i4"), ("s", "<i4"), ("rN", "<i4"), ("rj", "<i4"), ("ld", "<i4"),
                        ("off", "<i8")], align=False)
SECTOR_DT = np.dtype([("N", "<i4"), ("j", "<i4"), ("count", "<i4")], align=False)
ENV_BLOCK_DT = np.dtype([("aN", "<i4"), ("aj", "<i4"), ("w", "<i4"), ("bN", "<i4"), ("bj", "<i4"), ("rows", "<i4"),
                         ("cols", "<i4"), ("pad", "<i4"), ("off", "<i8")], align=False)
BOND_STATS_DT = np.dtype([("bond", "<i4"), ("direction", "<i4"), ("n_matvec", "<i4"), ("jacobi_sweeps", "<i4"),
                          ("chi_full", "<i4"), ("multiplets", "<i4"), ("n_tiles", "<i4"), ("n_segs", "<i4"),
                          ("theta_size", "<i8"), ("apply_flops", "<i8"), ("apply_bytes", "<i8"), ("svd_flops", "<i8"),
                          ("energy", "<f8"), ("residual", "<f8"), ("trunc_weight", "<f8"), ("t_plan", "<f8"),
                          ("t_lanczos", "<f8"), ("t_svd", "<f8"), ("t_env", "<f8"), ("t_total", "<f8"),
                          ("matvec_ms", "<f8")], align=False)
assert SITE_OP_DT.itemsize == 136 and MPO_ENTRY_DT.itemsize == 32 and SUBBLOCK_DT.itemsize == 32
assert SECTOR_DT.itemsize == 12 and ENV_BLOCK_DT.itemsize == 40 and BOND_STATS_DT.itemsize == 136


class Symmetry(C.Structure):
    """htn_symmetry"""
    _fields_ = [("kind", C.c_int32), ("n_site", C.c_int32), ("site_N", C.c_int32 * MAX_SITE),
                ("site_j", C.c_int32 * MAX_SITE)]


class SvdOpts(C.Structure):
    """htn_svd_opts"""
    _fields_ = [("split_elems", C.c_int32), ("sweeps_hint", C.c_int32), ("rank_cut", C.c_double),
                ("sweeps_used", C.POINTER(C.c_int32))]


class SweepOpts(C.Structure):
    """htn_sweep_opts"""
    _fields_ = [("chi_full", C.c_int32), ("weighting", C.c_int32), ("cutoff", C.c_double), ("krylovdim", C.c_int32),
                ("maxrestart", C.c_int32), ("lanczos_tol", C.c_double), ("jacobi_tol", C.c_double),
                ("jacobi_max_sweeps", C.c_int32), ("svd_split_elems", C.c_int32), ("rank_cut", C.c_double),
                ("profile", C.c_int32), ("pad", C.c_int32)]


EXPORTS = ["htn_last_error", "htn_abi_version", "htn_device_init", "htn_grouped_gemm_z",
           "htn_dots_scratch_elems", "htn_dots_z", "htn_axpys_z", "htn_scale_inv_sqrt_z",
           "htn_jacobi_svd_z", "htn_batched_copy_z", "htn_lanczos_scratch_elems", "htn_lanczos_z"]
# entry points shared by libhubbardtn_hip.so and the CPU baseline library (oracle/cpu_backend)
ENGINE_EXPORTS = ["htn_ctx_create", "htn_ctx_destroy", "htn_ctx_backend", "htn_ctx_set_timing", "htn_comm_unique_id",
                  "htn_ctx_set_comm", "htn_ctx_set_exchange", "htn_mpo_create", "htn_mpo_destroy", "htn_mps_create",
                  "htn_mps_destroy", "htn_bond_update", "htn_dmrg2_sweep", "htn_mps_theta_size", "htn_heff2_apply",
                  "htn_mps_get_theta", "htn_mps_nsites", "htn_mps_bond", "htn_mps_spectrum", "htn_mps_site_size",
                  "htn_mps_get_site", "htn_mps_env_size", "htn_mps_get_env", "htn_mps_env_blocks", "htn_mps_env_bond",
                  "htn_plan_apply_dump", "htn_mps_cache_stats", "htn_balance_tiles"]


class GemmLaunch(C.Structure):
    """htn_gemm_launch"""
    _fields_ = [("bufs", C.c_void_p * HTN_MAX_BUFS), ("tiles", C.c_void_p), ("segs", C.c_void_p),
                ("n_tiles", C.c_int32), ("pad", C.c_int32)]


EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_void_p)      # htn_exchange2_fn: != 0 aborts the solve

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
    lib.htn_jacobi_svd_z.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, f64, vp, C.POINTER(SvdOpts), vp]
    lib.htn_batched_copy_z.argtypes = [vp, vp, vp, vp, vp, i32, f64, vp]
    lib.htn_lanczos_scratch_elems.argtypes = [i32]
    lib.htn_lanczos_scratch_elems.restype = i64
    lib.htn_lanczos_z.argtypes = [C.POINTER(GemmLaunch), i32, i32, i32, vp, i64, i32, f64, i32, vp, i32,
                                  EXCHANGE_FN, vp, C.POINTER(f64), C.POINTER(i32), C.POINTER(f64), C.POINTER(f64), vp]
    declare_engine(lib)
    for name in EXPORTS + ENGINE_EXPORTS:
        getattr(lib, name)          # raises AttributeError if a declared symbol is missing
    if lib.htn_abi_version() != ABI_VERSION:
        raise HtnError("ABI version mismatch")
    if path is None:
        _lib = lib
    return lib


ABI_VERSION = 2


def declare_engine(lib):
    """prototypes of the bond-update / sweep level (same in the HIP library and in the CPU baseline library)"""
    vp, i32, i64, f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_double
    lib.htn_last_error.restype = C.c_char_p
    lib.htn_abi_version.restype = C.c_int
    lib.htn_ctx_create.argtypes = [i32, i32, vp, C.POINTER(vp)]
    lib.htn_ctx_destroy.argtypes = [vp]
    lib.htn_ctx_destroy.restype = None
    lib.htn_ctx_backend.argtypes = [vp]
    lib.htn_ctx_set_timing.argtypes = [vp, i32]
    lib.htn_comm_unique_id.argtypes = [vp]
    lib.htn_ctx_set_comm.argtypes = [vp, i32, i32, vp]
    lib.htn_ctx_set_exchange.argtypes = [vp, i32, i32, EXCHANGE_FN, vp]
    lib.htn_mpo_create.argtypes = [vp, C.POINTER(Symmetry), i32, vp, i32, vp, vp, vp, vp, C.POINTER(vp)]
    lib.htn_mpo_destroy.argtypes = [vp]
    lib.htn_mpo_destroy.restype = None
    lib.htn_mps_create.argtypes = [vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, C.POINTER(vp)]
    lib.htn_mps_destroy.argtypes = [vp]
    lib.htn_mps_destroy.restype = None
    lib.htn_bond_update.argtypes = [vp, i32, i32, i32, i32, C.POINTER(SweepOpts), vp]
    lib.htn_dmrg2_sweep.argtypes = [vp, C.POINTER(SweepOpts), vp, C.POINTER(f64)]
    lib.htn_mps_theta_size.argtypes = [vp, i32]
    lib.htn_mps_theta_size.restype = i64
    lib.htn_heff2_apply.argtypes = [vp, i32, vp, vp]
    lib.htn_mps_get_theta.argtypes = [vp, i32, vp]
    lib.htn_mps_nsites.argtypes = [vp]
    lib.htn_mps_bond.argtypes = [vp, i32, vp]
    lib.htn_mps_spectrum.argtypes = [vp, i32, vp, vp]
    lib.htn_mps_spectrum.restype = i64
    lib.htn_mps_site_size.argtypes = [vp, i32, C.POINTER(i32)]
    lib.htn_mps_site_size.restype = i64
    lib.htn_mps_get_site.argtypes = [vp, i32, vp, vp]
    lib.htn_mps_env_size.argtypes = [vp, i32, i32]
    lib.htn_mps_env_size.restype = i64
    lib.htn_mps_get_env.argtypes = [vp, i32, i32, vp]
    lib.htn_mps_env_blocks.argtypes = [vp, i32, i32, vp]
    lib.htn_mps_env_bond.argtypes = [vp, i32, i32, vp]
    lib.htn_plan_apply_dump.argtypes = [vp, i32, i32, C.POINTER(i32), vp, C.POINTER(i32), vp, C.POINTER(i64),
                                        C.POINTER(i64)]
    lib.htn_mps_cache_stats.argtypes = [vp, C.POINTER(i64), C.POINTER(i64)]
    lib.htn_balance_tiles.argtypes = [vp, i32, i32, vp, i32, C.POINTER(i32)]
    for name in ENGINE_EXPORTS:
        getattr(lib, name)


def check(lib, rc: int, what: str):
    if rc != 0:
        raise HtnError(f"{what}: {lib.htn_last_error().decode()}")
