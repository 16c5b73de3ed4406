"""TEST INFRASTRUCTURE: context provider for the CPU baseline library (oracle/cpu_backend/libhubbardtn_cpu.so).

Same C ABI and the same C++ planner / sweep driver as the product, host kernels instead of HIP: lets the
`-m "not gpu"` suite run the product's host logic end to end.  Never imported by hubbardtn_amd.
"""
import ctypes as C
import os

from hubbardtn_amd import abi
from oracle.cpu_backend import build as cpu_build


class CpuOps:
    name = "cpu-baseline"

    def __init__(self, lapack=False):
        if lapack:
            p = cpu_build.lapack_path()
            if p:
                os.environ.setdefault("HTN_CPU_LAPACK", p)
        self.lib = C.CDLL(cpu_build.build_library(verbose=False))
        abi.declare_engine(self.lib)
        if self.lib.htn_abi_version() != abi.ABI_VERSION:
            raise abi.HtnError("ABI version mismatch (CPU baseline)")
        h = C.c_void_p()
        abi.check(self.lib, self.lib.htn_ctx_create(abi.BACKEND_CPU, 0, None, C.byref(h)), "htn_ctx_create")
        self.ctx = h
        self._cb = None
        # small test problems: a few OpenMP threads are plenty, and a wide team spinning at barriers thrashes when the
        # test machine runs other jobs (bench.py's cpu_baseline sets the team to the host's core share itself)
        self.lib.htn_cpu_set_threads(min(4, os.cpu_count() or 1))

    def set_threads(self, n):
        """OpenMP team size of the host kernels; returns the previous setting"""
        return int(self.lib.htn_cpu_set_threads(int(n)))

    def set_exchange(self, rank, world, fn):
        """fn(y: numpy complex128 view) reduces y in place over the ranks; exceptions abort the solve and re-raise"""
        import numpy as np
        self._exc = None

        def _cb(y_ptr, n, user):
            try:
                buf = (C.c_double * (2 * n)).from_address(y_ptr)
                fn(np.frombuffer(buf, dtype=np.complex128))
                return 0
            except BaseException as e:      # noqa: BLE001 -- must not propagate through the C frame
                self._exc = e
                return 1
        self._cb = abi.EXCHANGE_FN(_cb) if fn is not None else abi.EXCHANGE_FN()
        abi.check(self.lib, self.lib.htn_ctx_set_exchange(self.ctx, rank, world, self._cb, None), "htn_ctx_set_exchange")

    def check_exchange(self):
        """re-raise (once) what the exchange hook raised during the library call that has just returned"""
        e, self._exc = getattr(self, "_exc", None), None
        if e is not None:
            raise e

    def __del__(self):
        h, self.ctx = getattr(self, "ctx", None), None
        if h:
            self.lib.htn_ctx_destroy(h)
