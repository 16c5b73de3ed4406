"""micro-benchmark of the H_eff apply (k_grouped_gemm_z) on the SURVEY App. D proxy sector table, scaled.
Task lists come from the Python statement of the planner (tests/ref_planner.py); `balance` applies the library's launch
balancing pass (htn_balance_tiles: tile cuts + placement-aware order; knobs HTN_GEMM_CAP / HTN_GEMM_CAPK / HTN_GEMM_XCD / HTN_GEMM_LAYERS)."""
import os
import sys
import time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import ref_planner as pl
from hubbardtn_amd import models
from hubbardtn_amd.device import HipOps

TAB = {(2, 0): 2, (2, 2): 1, (3, 1): 12, (3, 3): 3, (4, 0): 23, (4, 2): 24, (4, 4): 4, (5, 1): 53, (5, 3): 24, (5, 5): 3,
       (6, 0): 44, (6, 2): 52, (6, 4): 13, (6, 6): 1, (7, 1): 53, (7, 3): 24, (7, 5): 3, (8, 0): 23, (8, 2): 24, (8, 4): 4,
       (9, 1): 12, (9, 3): 3, (10, 0): 2, (10, 2): 1}


def run(ops, scale, t=(1.0,), reps=30, balance=True, evict=False):
    L = 64
    mpo = models.hamiltonian(models.OB_Sim(list(t), [4.0]), L)
    bl = pl.Bond({(N + 25, j): max(1, int(round(n * scale))) for (N, j), n in TAB.items()})
    br = pl.Bond({(N + 27, j): max(1, int(round(n * scale))) for (N, j), n in TAB.items()})
    tl = pl.ThetaLayout.build(bl, br)
    Ll = pl.EnvLayout.build("L", bl, mpo[31].left)
    Rl = pl.EnvLayout.build("R", br, mpo[32].right)
    tz, ty, zs, nt = pl.plan_apply(tl, Ll, Rl, mpo[31], mpo[32])
    rng = np.random.default_rng(0)
    rz = lambda n: ops.to_device(rng.standard_normal(n) + 1j * rng.standard_normal(n))
    x, y, Lb, Rb, z = rz(tl.size), ops.zeros_z(tl.size), rz(max(Ll.size, 1)), rz(max(Rl.size, 1)), ops.zeros_z(max(zs, 1))
    bufs = [None] * 8
    bufs[pl.BUF_X], bufs[pl.BUF_Y], bufs[pl.BUF_L], bufs[pl.BUF_R], bufs[pl.BUF_Z] = x, y, Lb, Rb, z
    dz = ops.upload_tasks(tz, balance=balance) if tz is not None else None
    dy = ops.upload_tasks(ty, balance=balance)
    junk = ops.zeros_z(40_000_000) if evict else None      # 640 MB streamed between applies: operands leave L2 and MALL

    def go():
        if junk is not None:
            junk.add_(1.0)
        if dz is not None:
            ops.grouped_gemm(bufs, dz)
        ops.grouped_gemm(bufs, dy)
    for _ in range(3):
        go()
    torch.cuda.synchronize()
    tot = 0.0
    for _ in range(reps):
        if junk is not None:
            junk.add_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if dz is not None:
            ops.grouped_gemm(bufs, dz)
        ops.grouped_gemm(bufs, dy)
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1) * 1e3
    us = tot / reps
    fl = ty.flops + (tz.flops if tz is not None else 0)
    print(f"scale {scale:4.1f} t={t} balance={int(balance)} evict={int(evict)} cap={os.environ.get('HTN_GEMM_CAP', 'auto')} "
          f"xcd={os.environ.get('HTN_GEMM_XCD', '1')}: chi_full {bl.dim_full:5d} |theta| {tl.size:8d} tiles {ty.ntiles:5d} -> {dy[1]:5d} "
          f"segs {ty.nsegs:6d} GFLOP {fl / 1e9:7.3f}  {us:8.1f} us  {fl / us / 1e6:7.2f} TFLOP/s", flush=True)


if __name__ == "__main__":
    ops = HipOps(0)
    scales = [float(v) for v in sys.argv[1:]] or [1.0, 2.0]
    for s in scales:
        run(ops, s, balance=False)
        run(ops, s, balance=True)
        run(ops, s, balance=True, evict=True)
