"""the C-ABI library loads and exports every symbol include/hubbardtn_hip.h declares (no compute calls)"""
import ctypes
import os
import re

import numpy as np

from hubbardtn_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "hubbardtn_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(htn_[a-z0-9_]+)\s*\(", src)) - {"htn_exchange2_fn"})


def test_library_exports_every_declared_symbol():
    lib = abi.load_library()
    names = _declared_functions()
    assert len(names) >= 12
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(abi.EXPORTS + abi.ENGINE_EXPORTS) == names
    assert lib.htn_abi_version() == 2
    assert lib.htn_last_error() == b""


def test_cpu_baseline_library_exports_the_engine_level_abi():
    """oracle/cpu_backend builds the same planner / sweep-driver sources against host kernels: every bond-update / sweep
    entry point of the header exists there too, and neither library contains the other's backend"""
    import ctypes as C
    from oracle.cpu_backend import build as cpu_build
    lib = C.CDLL(cpu_build.build_library(verbose=False))
    abi.declare_engine(lib)
    for n in abi.ENGINE_EXPORTS:
        assert hasattr(lib, n), n
    assert lib.htn_abi_version() == 2
    for n in ("htn_grouped_gemm_z", "htn_jacobi_svd_z", "htn_lanczos_z"):       # kernel-level entry points are HIP only
        assert not hasattr(lib, n)
    hip = abi.load_library()
    ctx = C.c_void_p()
    assert hip.htn_ctx_create(abi.BACKEND_CPU, 0, None, C.byref(ctx)) != 0      # the product has no CPU backend
    assert b"no CPU" in hip.htn_last_error()


def test_struct_layouts_match_header():
    assert abi.TILE_DT.itemsize == 64 and abi.SEG_DT.itemsize == 64
    assert abi.SVD_DT.itemsize == 40 and abi.COPY_DT.itemsize == 64
    assert ctypes.sizeof(abi.GemmLaunch) == 8 * 8 + 8 + 8 + 4 + 4
    assert ctypes.sizeof(abi.SvdOpts) == 24                 # htn_svd_opts
    assert abi.TILE_DT.fields["seg_begin"][1] == 32 and abi.SEG_DT.fields["alpha_re"][1] == 48


def test_product_has_no_cpu_path(monkeypatch):
    """the product's only device-ops provider needs the GPU: constructing it without one raises"""
    import torch
    from hubbardtn_amd.device import HipOps
    if not torch.cuda.is_available():
        try:
            HipOps(0)
        except abi.HtnError as e:
            assert "no CPU fallback" in str(e)
        else:
            raise AssertionError("HipOps must fail loudly without a GPU")
