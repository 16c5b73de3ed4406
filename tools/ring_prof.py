"""diagnostic: per-phase time of k_jacobi_ring on one block (needs a library built with HTN_EXTRA_FLAGS=-DHTN_RING_PROF)"""
import ctypes as C
import sys
import numpy as np
sys.path.insert(0, ".")
from hubbardtn_amd import abi
from hubbardtn_amd.device import HipOps

ops = HipOps(0)
rng = np.random.default_rng(0)
rz = lambda n: rng.standard_normal(n) + 1j * rng.standard_normal(n)
for a in sys.argv[1:]:
    m0, n0 = (int(x) for x in a.split("x"))
    r = min(m0, n0)
    U, _ = np.linalg.qr(rz(m0 * r).reshape(m0, r))
    W, _ = np.linalg.qr(rz(n0 * r).reshape(n0, r))
    M = (U * 10.0 ** (-12 * np.arange(r) / max(r - 1, 1))) @ W.conj().T
    desc = np.zeros(1, dtype=abi.SVD_DT)
    desc[0] = (0, 0, 0, n0, r, abi.SVD_QRCP, m0)
    G = ops.to_device(M.T.reshape(-1).copy())
    V = ops.zeros_z(((max(m0, n0) + 63) // 64 * 64) * max(m0, n0))
    S, info = ops.empty_f64(max(m0, n0)), ops.empty_i32(1)
    for _ in range(2):
        G.copy_(ops.to_device(M.T.reshape(-1).copy()))
        ops.jacobi_svd(G, V, S, ops.to_device(desc), 1, max(m0, n0), 40, 1e-14, info, desc_host=desc)
    if hasattr(ops.lib, "htn_qr_prof_dump"):
        q = np.zeros(8, dtype=np.int64)
        ops.lib.htn_qr_prof_dump(C.c_void_p(q.ctypes.data))
        print(f"{a}: k_qr_large us: window {q[0] / 100:.1f}  panel load {q[1] / 100:.1f}  panel steps {q[2] / 100:.1f}  flush {q[3] / 100:.1f}  "
              f"trailing {q[4] / 100:.1f}  total {q[5] / 100:.1f}  panels {q[6]}")
    out = np.zeros(256 * 8, dtype=np.int64)
    ops.lib.htn_ring_prof_dump(C.c_void_p(out.ctypes.data))
    out = out.reshape(256, 8)
    rows = [r for r in range(256) if out[r, 6] > 0]          # (grid positions with work: the XCD-aware placement leaves gaps)
    P = int(out[rows[0], 7] % 1000)
    sw = int(out[rows[0], 7] // 1000 % 1000)
    local = int(out[rows[0], 7] // 1000000)
    names = ["cross", "send+drain", "flags+wait", "recv", "intra", "conv", "total"]
    print(f"{a}: P={P} sweeps={sw} hand-off through one XCD's L2: {bool(local)}  grid positions {rows[:P]}  (us per sweep, per workgroup; 100 MHz ticks)")
    for k, r in enumerate(rows[:P]):
        print("  k=%2d " % k + "  ".join(f"{n} {out[r, q] / 100.0 / max(sw, 1):7.1f}" for q, n in enumerate(names)))
