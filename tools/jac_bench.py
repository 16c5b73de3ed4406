"""micro-benchmark of k_jacobi_svd on single graded blocks: time vs size / path"""
import sys
import numpy as np
sys.path.insert(0, ".")
import torch
from hubbardtn_amd import abi
from hubbardtn_amd.device import HipOps

ops = HipOps(0)
rng = np.random.default_rng(0)


def rz(n):
    return rng.standard_normal(n) + 1j * rng.standard_normal(n)


def run(m0, n0, flags, reps=5, multi=False):
    r = min(m0, n0)
    U, _ = np.linalg.qr(rz(m0 * r).reshape(m0, r))
    W, _ = np.linalg.qr(rz(n0 * r).reshape(n0, r))
    s = 10.0 ** (-12 * np.arange(r) / max(r - 1, 1))
    M = (U * s) @ W.conj().T
    desc = np.zeros(1, dtype=abi.SVD_DT)
    if flags & abi.SVD_QRCP:
        desc[0] = (0, 0, 0, n0, r, flags, m0)
    else:
        desc[0] = (0, 0, 0, m0, n0, flags, 0)
    d_desc = ops.to_device(desc)
    src = ops.to_device(M.T.reshape(-1).copy())
    G = ops.empty_z(m0 * n0)
    V = ops.zeros_z(((max(m0, n0) + 63) // 64 * 64) * max(m0, n0))
    S = ops.empty_f64(max(m0, n0))
    info = ops.empty_i32(1)
    ts = []
    for _ in range(reps):
        G.copy_(src)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.jacobi_svd(G, V, S, d_desc, 1, max(m0, n0), 40, 1e-14, info, desc_host=desc if multi else None)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return min(ts), int(info.cpu()[0])


sizes = [(48, 48), (64, 64), (80, 80), (96, 96), (107, 107), (128, 128), (160, 160), (202, 202), (256, 256), (400, 400)]
if len(sys.argv) > 1:       # e.g. `jac_bench.py 256x256 202x180`: multi-launch path only
    for a in sys.argv[1:]:
        m0, n0 = (int(x) for x in a.split("x"))
        c = run(m0, n0, abi.SVD_QRCP, multi=True)
        print(f"{m0}x{n0}: multi-launch {c[0]:.3f} ms ({c[1]} sweeps)", flush=True)
    sys.exit(0)
for (m0, n0) in sizes:
    a = run(m0, n0, abi.SVD_QRCP)
    c = run(m0, n0, abi.SVD_QRCP, multi=True)
    print(f"{m0}x{n0}: qrcp one-CU {a[0]:.3f} ms ({a[1]} sweeps)   multi-launch {c[0]:.3f} ms ({c[1]} sweeps)", flush=True)
