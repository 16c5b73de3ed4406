"""micro-benchmark of the H_eff apply (k_grouped_gemm_z) on the SURVEY App. D proxy sector table, scaled"""
import sys
import time
import numpy as np
sys.path.insert(0, ".")
import torch
from hubbardtn_amd import models, planner as pl
from hubbardtn_amd.device import HipOps

TAB = {(2, 0): 2, (2, 2): 1, (3, 1): 12, (3, 3): 3, (4, 0): 23, (4, 2): 24, (4, 4): 4, (5, 1): 53, (5, 3): 24, (5, 5): 3,
       (6, 0): 44, (6, 2): 52, (6, 4): 13, (6, 6): 1, (7, 1): 53, (7, 3): 24, (7, 5): 3, (8, 0): 23, (8, 2): 24, (8, 4): 4,
       (9, 1): 12, (9, 3): 3, (10, 0): 2, (10, 2): 1}


def run(ops, scale, t=(1.0,), reps=30):
    L = 64
    mpo = models.hamiltonian(models.OB_Sim(list(t), [4.0]), L)
    bl = pl.Bond({(N + 25, j): max(1, int(round(n * scale))) for (N, j), n in TAB.items()})
    br = pl.Bond({(N + 27, j): max(1, int(round(n * scale))) for (N, j), n in TAB.items()})
    tl = pl.ThetaLayout.build(bl, br)
    Ll = pl.EnvLayout.build("L", bl, mpo[31].left)
    Rl = pl.EnvLayout.build("R", br, mpo[32].right)
    t0 = time.time()
    tz, ty, zs, nt = pl.plan_apply(tl, Ll, Rl, mpo[31], mpo[32])
    tplan = time.time() - t0
    rng = np.random.default_rng(0)
    rz = lambda n: ops.to_device(rng.standard_normal(n) + 1j * rng.standard_normal(n))
    x, y, Lb, Rb, z = rz(tl.size), ops.zeros_z(tl.size), rz(max(Ll.size, 1)), rz(max(Rl.size, 1)), ops.zeros_z(max(zs, 1))
    bufs = [None] * 8
    bufs[pl.BUF_X], bufs[pl.BUF_Y], bufs[pl.BUF_L], bufs[pl.BUF_R], bufs[pl.BUF_Z] = x, y, Lb, Rb, z
    dz = ops.upload_tasks(tz) if tz is not None else None
    dy = ops.upload_tasks(ty)

    def go():
        if dz is not None:
            ops.grouped_gemm(bufs, dz)
        ops.grouped_gemm(bufs, dy)
    for _ in range(3):
        go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        go()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    fl = ty.flops + (tz.flops if tz is not None else 0)
    print(f"scale {scale:4.1f} t={t}: chi_full {bl.dim_full:5d} mult {bl.multiplets:4d} |theta| {tl.size:8d} tiles {ty.ntiles:5d} "
          f"segs {ty.nsegs:6d} GFLOP {fl / 1e9:7.3f}  {us:8.1f} us  {fl / us / 1e6:7.2f} TFLOP/s  plan {tplan:.2f}s", flush=True)


if __name__ == "__main__":
    ops = HipOps(0)
    for s in (0.5, 1.0, 2.0, 4.0):
        run(ops, s)
    run(ops, 1.0, t=(1.0, 0.1))
