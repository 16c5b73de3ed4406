"""Builds libhubbardtn_hip.so in-tree with hipcc for gfx950 (the only supported target)."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRCS = [os.path.join(HERE, "csrc", f) for f in ("htn_abi.hip", "htn_gemm.hip", "htn_krylov.hip", "htn_svd.hip",
                                                "htn_backend_hip.hip", "htn_plan.cpp", "htn_engine.cpp")]
LIB = os.path.join(HERE, "csrc", "libhubbardtn_hip.so")
INC = os.path.join(ROOT, "include")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = SRCS + [os.path.join(HERE, "csrc", "htn_common.h"), os.path.join(HERE, "csrc", "htn_core.h"),
                   os.path.join(INC, "hubbardtn_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # -amdgpu-mfma-vgpr-form: keep MFMA accumulators in VGPRs.  Without it the 256-thread kernels get AGPR
    # accumulators whose loop-carried values are copied VGPR<->AGPR around every MFMA group (48 moves + a full
    # pipeline drain per k-step in k_jacobi_pairs_gram: 200 instead of 64 cycles per MFMA).
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-mllvm", "-amdgpu-mfma-vgpr-form"] \
        + os.environ.get("HTN_EXTRA_FLAGS", "").split() \
        + ["-std=c++17", "-fPIC", "-shared", "-I", INC, "-I", os.path.join(HERE, "csrc"), *SRCS, "-o", LIB, "-ldl"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)
