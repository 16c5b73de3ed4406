"""Context provider of the product path + thin wrappers over the kernel-level C ABI (hubbardtn_amd/abi.py).

`HipOps.ctx` is the `htn_ctx` every engine object (hubbardtn_amd/engine.py) lives in: the library owns its stream,
its device memory and -- for the sector-parallel apply -- its RCCL communicator (`set_comm`).  PyTorch is plumbing only:
`torch.distributed` distributes the 128-byte RCCL id, and the kernel-level wrappers below (used by the kernel tests)
take torch tensors as device memory.  Every compute call goes through libhubbardtn_hip.so; if the library or the GPU
is missing, construction raises -- there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import abi


class HipOps:
    """the one and only implementation of the device primitives used by the product path"""

    name = "hip"

    def __init__(self, device: int = 0):
        import torch
        self.torch = torch
        self.lib = abi.load_library()
        if not torch.cuda.is_available():
            raise abi.HtnError("no HIP device visible: hubbardtn_amd has no CPU fallback")
        torch.cuda.set_device(device)
        self.device = torch.device("cuda", device)
        name = C.create_string_buffer(256)
        cus = C.c_int(0)
        abi.check(self.lib, self.lib.htn_device_init(device, name, C.byref(cus)), "htn_device_init")
        self.arch = name.value.decode()
        self.cu_count = cus.value
        self._dots_scratch = None
        self._ws = None              # split-K workspace of the kernel-level grouped_gemm wrapper (tickets | slabs)
        self._stream_ptr = None
        self.event_log = None        # bench.py: list of (start_event, end_event, tag, flops)
        h = C.c_void_p()
        abi.check(self.lib, self.lib.htn_ctx_create(abi.BACKEND_HIP, device, None, C.byref(h)), "htn_ctx_create")
        self.ctx = h
        self._cb = None
        self._exc = None

    def __del__(self):
        h, self.ctx = getattr(self, "ctx", None), None
        if h:
            self.lib.htn_ctx_destroy(h)

    # ---- context settings ---------------------------------------------------------------------
    def set_timing(self, on: bool):
        """HIP events around every H_eff apply launch (BondStats.matvec_ms)"""
        abi.check(self.lib, self.lib.htn_ctx_set_timing(self.ctx, 1 if on else 0), "htn_ctx_set_timing")

    def set_comm(self, rank: int, world: int, group=None):
        """sector-parallel apply over `world` ranks: RCCL all-reduce inside the library.  The id is created on rank 0
        and broadcast through torch.distributed (any initialised backend)."""
        import torch.distributed as dist
        ident = (C.c_char * 128)()
        if rank == 0:
            abi.check(self.lib, self.lib.htn_comm_unique_id(ident), "htn_comm_unique_id")
        if world > 1:
            box = [bytes(ident)]
            dist.broadcast_object_list(box, src=0, group=group)
            ident = (C.c_char * 128).from_buffer_copy(box[0])
        abi.check(self.lib, self.lib.htn_ctx_set_comm(self.ctx, rank, world, ident), "htn_ctx_set_comm")

    def set_exchange(self, rank: int, world: int, fn):
        """caller-supplied reduction: fn(y_ptr, n) must enqueue/perform the sum of y (n complex128 at device pointer
        y_ptr) over the ranks.  An exception inside fn cannot cross the C frame: it is stored, the C side sees a non-zero
        return and aborts the solve, and the engine call that was running re-raises the ORIGINAL exception (with its
        traceback) through `check_exchange` as soon as the library call has returned -- KeyboardInterrupt / SystemExit
        included, so a rank that must die does not carry on towards the next collective."""
        def _cb(y_ptr, n, user):
            try:
                fn(y_ptr, n)
                return 0
            except BaseException as e:      # noqa: BLE001 -- stored and re-raised by check_exchange after the C call
                self._exc = e
                return 1
        self._cb = abi.EXCHANGE_FN(_cb) if fn is not None else abi.EXCHANGE_FN()
        abi.check(self.lib, self.lib.htn_ctx_set_exchange(self.ctx, rank, world, self._cb, None), "htn_ctx_set_exchange")

    def check_exchange(self):
        """re-raise (once) what the exchange hook raised during the library call that has just returned"""
        e, self._exc = self._exc, None
        if e is not None:
            raise e

    # ---- memory -----------------------------------------------------------------------------
    def empty_z(self, n):
        return self.torch.empty(int(n), dtype=self.torch.complex128, device=self.device)

    def zeros_z(self, n):
        return self.torch.zeros(int(n), dtype=self.torch.complex128, device=self.device)

    def empty_f64(self, n):
        return self.torch.empty(int(n), dtype=self.torch.float64, device=self.device)

    def empty_i32(self, n):
        return self.torch.empty(int(n), dtype=self.torch.int32, device=self.device)

    def to_device(self, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        if arr.dtype.fields is not None:            # struct arrays travel as bytes
            arr = arr.view(np.uint8)
        # a pageable-memory .to(device) returns after the source has been staged: no defensive copy needed
        return self.torch.from_numpy(arr).to(self.device)

    def to_device_packed(self, arrays):
        """several small host arrays -> ONE host-to-device copy; returns device views (one per array, 16-byte
        aligned inside the packed buffer).  The per-bond bookkeeping arrays are a few hundred bytes each and a
        separate copy costs ~20 us of host time apiece."""
        raws = [np.ascontiguousarray(a).view(np.uint8).reshape(-1) for a in arrays]
        offs, pos = [], 0
        for r in raws:
            offs.append(pos)
            pos += (r.size + 15) // 16 * 16
        buf = np.zeros(max(pos, 16), dtype=np.uint8)
        for r, o in zip(raws, offs):
            buf[o:o + r.size] = r
        dev = self.torch.from_numpy(buf).to(self.device)
        out = []
        for a, r, o in zip(arrays, raws, offs):
            v = dev[o:o + max(r.size, 1)]
            a = np.asarray(a)
            if a.dtype.fields is None and a.dtype != np.uint8:
                v = v.view(getattr(self.torch, {"int32": "int32", "int64": "int64", "float64": "float64",
                                                 "complex128": "complex128"}[a.dtype.name])) if r.size else v
            out.append(v)
        return out

    def to_host(self, t) -> np.ndarray:
        return t.cpu().numpy()

    def copy_(self, dst, src):
        dst.copy_(src)

    def zero(self, t):
        t.zero_()

    def scale_inplace(self, t, f: float):
        t.mul_(f)

    def sync(self):
        self.torch.cuda.current_stream().synchronize()

    def _stream(self):
        # the launch stream is looked up once (torch.cuda.current_stream() costs ~10 us per call, ~1000 calls per
        # sweep); use_stream() re-binds when the caller switches torch's current stream
        s = self._stream_ptr
        if s is None:
            s = self._stream_ptr = C.c_void_p(self.torch.cuda.current_stream().cuda_stream)
        return s

    def use_stream(self):
        """re-read torch's current stream (call after torch.cuda.set_stream / inside a torch.cuda.stream block)"""
        self._stream_ptr = None

    @staticmethod
    def _p(t):
        return C.c_void_p(0 if t is None else t.data_ptr())

    # ---- kernels ----------------------------------------------------------------------------
    def upload_tasks(self, tasks, balance=False):
        """tasks: Tasks (tests/ref_planner.py) -> (tiles_dev, ntiles, segs_dev).  balance=True applies the library's own
        launch balancing pass (htn_balance_tiles: spatial and split-K tile cuts + placement-aware order) and makes sure the
        split-K workspace is large enough; grouped_gemm passes the workspace in buffer slot 7."""
        tiles_h, ntiles = tasks.tiles, tasks.ntiles
        if balance and ntiles > 0:
            n_out = C.c_int32(0)
            src = np.ascontiguousarray(tasks.tiles[:ntiles])
            self.lib.htn_balance_tiles(src.ctypes.data, ntiles, self.cu_count, None, 0, C.byref(n_out))
            out = np.zeros(max(n_out.value, 1), dtype=abi.TILE_DT)
            slots = self.lib.htn_balance_tiles(src.ctypes.data, ntiles, self.cu_count, out.ctypes.data, len(out), C.byref(n_out))
            if slots < 0:
                raise abi.HtnError(self.lib.htn_last_error().decode())
            tiles_h, ntiles = out, n_out.value
            need = abi.WS_TICKET_ELEMS + slots * abi.HTN_TILE * abi.HTN_TILE
            if self._ws is None or self._ws.numel() < need:
                self._ws = self.zeros_z(2 * need)
        tiles, segs = self.to_device_packed([tiles_h, tasks.segs])
        return (tiles, ntiles, segs)

    def grouped_gemm(self, bufs, dev_tasks, tag=None, flops=0):
        tiles, ntiles, segs = dev_tasks
        if ntiles == 0:
            return
        table = (C.c_void_p * abi.HTN_MAX_BUFS)(*[0 if b is None else b.data_ptr() for b in bufs])
        if self._ws is not None and bufs[abi.BUF_WS] is None:
            table[abi.BUF_WS] = self._ws.data_ptr()
        log = self.event_log is not None and tag is not None
        if log:      # HIP events on the launch stream (torch's current stream IS the launch stream)
            e0 = self.torch.cuda.Event(enable_timing=True)
            e1 = self.torch.cuda.Event(enable_timing=True)
            e0.record()
        abi.check(self.lib, self.lib.htn_grouped_gemm_z(table, self._p(tiles), ntiles, self._p(segs),
                                                        self._stream()), "htn_grouped_gemm_z")
        if log:
            e1.record()
            self.event_log.append((e0, e1, tag, flops))

    def dots(self, V, ldv, nvec, w, n, out):
        need = self.lib.htn_dots_scratch_elems(64)
        if self._dots_scratch is None or self._dots_scratch.numel() < need:
            self._dots_scratch = self.empty_z(need)
        assert nvec <= 64
        abi.check(self.lib, self.lib.htn_dots_z(self._p(V), ldv, nvec, self._p(w), n, self._p(out),
                                                self._p(self._dots_scratch), self._stream()), "htn_dots_z")

    def axpys(self, w, V, ldv, nvec, coef, sign, n):
        abi.check(self.lib, self.lib.htn_axpys_z(self._p(w), self._p(V), ldv, nvec, self._p(coef), float(sign),
                                                 n, self._stream()), "htn_axpys_z")

    def scale_inv_sqrt(self, dst, src, nrm2, n):
        abi.check(self.lib, self.lib.htn_scale_inv_sqrt_z(self._p(dst), self._p(src), self._p(nrm2), n,
                                                          self._stream()), "htn_scale_inv_sqrt_z")

    def lanczos(self, stages, x_slot, y_slot, V, n, krylovdim, tol, max_restart, zero_y=False, exchange=None):
        """lowest eigenpair of the map defined by `stages` = [(bufs, dev_tasks), ...]; V[0:n] start -> result.
        exchange(y_tensor) is called after every matvec has been enqueued (multi-GPU all-reduce hook).
        Returns (eigenvalue, n_matvec, residual)."""
        arr = (abi.GemmLaunch * len(stages))()
        keep = []
        for k, (bufs, (tiles, ntiles, segs)) in enumerate(stages):
            for b in range(abi.HTN_MAX_BUFS):
                arr[k].bufs[b] = 0 if bufs[b] is None else bufs[b].data_ptr()
            arr[k].tiles, arr[k].segs, arr[k].n_tiles = tiles.data_ptr(), segs.data_ptr(), ntiles
            keep.append((bufs, tiles, segs))
        need = self.lib.htn_lanczos_scratch_elems(krylovdim)
        if getattr(self, "_lan_scratch", None) is None or self._lan_scratch.numel() < need:
            self._lan_scratch = self.empty_z(need)
        base = V.data_ptr()

        err = []

        def _cb(y_ptr, nn, user):
            try:
                off = (y_ptr - base) // 16
                exchange(V[off:off + nn])
                return 0
            except BaseException as e:      # noqa: BLE001 -- a Python exception must not cross the C frame
                err.append(e)
                return 1
        cb = abi.EXCHANGE_FN(_cb) if exchange is not None else abi.EXCHANGE_FN()
        eig, nmv, res, ms = C.c_double(0.0), C.c_int32(0), C.c_double(0.0), C.c_double(0.0)
        timed = self.event_log is not None
        rc = self.lib.htn_lanczos_z(arr, len(stages), x_slot, y_slot, self._p(V), n, krylovdim,
                                    float(tol), max_restart, self._p(self._lan_scratch),
                                    1 if zero_y else 0, cb, None, C.byref(eig), C.byref(nmv),
                                    C.byref(res), C.byref(ms) if timed else None, self._stream())
        if err:
            raise err[0]
        abi.check(self.lib, rc, "htn_lanczos_z")
        if timed:
            self.event_log.append(("matvec_ms", ms.value, nmv.value))
        return eig.value, nmv.value, res.value

    def jacobi_svd(self, G, Vj, S, desc_dev, nblocks, max_m, max_sweeps, tol, info, desc_host=None, split=0, rank_cut=0.0,
                   sweeps_hint=0):
        """split: elements of R^H above which a block takes the large-block SVD path (0 = default);
        rank_cut: absolute singular-value cut of the large blocks' rank-revealing QR (0 = off); sweeps_hint: expected outer
        sweeps of the large-block path (bounds the speculative enqueue).  Per call (ABI 2).  Returns the outer sweeps the
        large-block path used (0: no block took it)."""
        hp = C.c_void_p(desc_host.ctypes.data) if desc_host is not None else C.c_void_p(0)
        used = C.c_int32(0)
        opts = abi.SvdOpts(int(split), int(sweeps_hint), float(rank_cut), C.pointer(used))
        abi.check(self.lib, self.lib.htn_jacobi_svd_z(self._p(G), self._p(Vj), self._p(S), self._p(desc_dev), hp,
                                                      nblocks, max_m, max_sweeps, float(tol), self._p(info),
                                                      C.byref(opts), self._stream()), "htn_jacobi_svd_z")
        return used.value

    def batched_copy(self, dst, src, idx, scl, items_dev, nitems, gscale):
        if nitems == 0:
            return
        abi.check(self.lib, self.lib.htn_batched_copy_z(self._p(dst), self._p(src), self._p(idx), self._p(scl),
                                                        self._p(items_dev), nitems, float(gscale),
                                                        self._stream()), "htn_batched_copy_z")
