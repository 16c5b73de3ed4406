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
    return sorted(set(re.findall(r"\b(htn_[a-z0-9_]+)\s*\(", src)) - {"htn_exchange_fn"})


def test_library_exports_every_declared_symbol():
    lib = abi.load_library()
    names = _declared_functions()
    assert len(names) >= 12
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(abi.EXPORTS) == names
    assert lib.htn_abi_version() == 1
    assert lib.htn_last_error() == b""


def test_struct_layouts_match_header():
    assert abi.TILE_DT.itemsize == 48 and abi.SEG_DT.itemsize == 64
    assert abi.SVD_DT.itemsize == 40 and abi.COPY_DT.itemsize == 64
    assert ctypes.sizeof(abi.GemmLaunch) == 8 * 8 + 8 + 8 + 4 + 4
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
