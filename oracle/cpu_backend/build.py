"""Builds oracle/cpu_backend/libhubbardtn_cpu.so: the CPU baseline / checker behind the same C ABI as the product
(see htn_backend_cpu.cpp).  g++ + OpenMP; the planner and sweep-driver sources are the product's own
(hubbardtn_amd/csrc/htn_plan.cpp, htn_engine.cpp), only the kernels differ."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "hubbardtn_amd", "csrc")
SRCS = [os.path.join(CSRC, "htn_plan.cpp"), os.path.join(CSRC, "htn_engine.cpp"), os.path.join(HERE, "htn_backend_cpu.cpp")]
LIB = os.path.join(HERE, "libhubbardtn_cpu.so")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = SRCS + [os.path.join(CSRC, "htn_core.h"), os.path.join(ROOT, "include", "hubbardtn_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB
    cmd = [os.environ.get("CXX", "g++"), "-O3", "-march=x86-64-v3", "-std=c++17", "-fPIC", "-shared", "-fopenmp",
           "-I", os.path.join(ROOT, "include"), "-I", CSRC, *SRCS, "-o", LIB, "-ldl"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return LIB


def lapack_path():
    """an OpenBLAS with LAPACKE bundled with scipy / numpy (the only BLAS in the image), or None"""
    import glob
    for pkg in ("scipy", "numpy"):
        try:
            mod = __import__(pkg)
        except Exception:
            continue
        base = os.path.join(os.path.dirname(os.path.dirname(mod.__file__)), pkg + ".libs")
        for p in sorted(glob.glob(os.path.join(base, "libscipy_openblas*.so"))):
            return p
    return None


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)
