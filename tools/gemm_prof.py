"""diagnostic: per-workgroup timeline of one H_eff apply launch (needs a library built with -DHTN_GEMM_PROF)"""
import ctypes as C
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import ref_planner as pl
from hubbardtn_amd import models
from hubbardtn_amd.device import HipOps
import apply_bench as ab

ops = HipOps(0)
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
balance = int(sys.argv[2]) if len(sys.argv) > 2 else 1
L = 64
mpo = models.hamiltonian(models.OB_Sim([1.0], [4.0]), L)
bl = pl.Bond({(N + 25, j): max(1, int(round(n * scale))) for (N, j), n in ab.TAB.items()})
br = pl.Bond({(N + 27, j): max(1, int(round(n * scale))) for (N, j), n in ab.TAB.items()})
tl = pl.ThetaLayout.build(bl, br)
Ll = pl.EnvLayout.build("L", bl, mpo[31].left)
Rl = pl.EnvLayout.build("R", br, mpo[32].right)
tz, ty, zs, nt = pl.plan_apply(tl, Ll, Rl, mpo[31], mpo[32])
rng = np.random.default_rng(0)
rz = lambda n: ops.to_device(rng.standard_normal(n) + 1j * rng.standard_normal(n))
x, y, Lb, Rb = rz(tl.size), ops.zeros_z(tl.size), rz(max(Ll.size, 1)), rz(max(Rl.size, 1))
bufs = [None] * 8
bufs[pl.BUF_X], bufs[pl.BUF_Y], bufs[pl.BUF_L], bufs[pl.BUF_R], bufs[pl.BUF_Z] = x, y, Lb, Rb, ops.zeros_z(max(zs, 1))
dy = ops.upload_tasks(ty, balance=bool(balance))
for _ in range(20):
    ops.grouped_gemm(bufs, dy)
torch.cuda.synchronize()
n = dy[1]
out = np.zeros((8192, 6), dtype=np.int64)
ops.lib.htn_gemm_prof_dump.argtypes = [C.c_void_p, C.c_int]
assert ops.lib.htn_gemm_prof_dump(out.ctypes.data, n) == 0
out = out[:n]
tiles = ops.to_host(dy[0]).view(np.uint8).reshape(-1)[:n * 64].view(ab.pl.abi.TILE_DT) if hasattr(ab.pl, "abi") else None
t0 = out[:, 0].min()
st, kl, en = (out[:, 0] - t0) / 100.0, (out[:, 1] - t0) / 100.0, (out[:, 2] - t0) / 100.0
hw, xcc = out[:, 3] & 0xffffffff, out[:, 3] >> 32
cu = (hw >> 8) & 0xf
se = (hw >> 13) & 0x7
cuid = xcc * 64 + se * 16 + cu
print(f"tiles {n}: kernel span {en.max():.1f} us; start: median {np.median(st):.1f} max {st.max():.1f}; duration median {np.median(en - st):.1f} max {(en - st).max():.1f}")
print("distinct CUs", len(np.unique(cuid)), "XCC histogram", np.bincount(xcc.astype(int)))
order = np.argsort(-(en - st))
cyc = out[:, 4]
nseg, nq = out[:, 5] >> 8, out[:, 5] & 0xff
ghz = cyc / np.maximum((kl - st) * 1e3, 1e-9)
print("in-kernel clock over the K loops (GHz): median %.2f min %.2f max %.2f" % (np.median(ghz), ghz.min(), ghz.max()))
for q in (1, 2, 4):
    sel = (nq == q) & (nseg > 0)
    if sel.any():
        per = (kl - st)[sel] / np.ceil(nseg[sel] * q / 4.0)
        print(f"  {q}-quadrant tiles: {sel.sum()}  K loop per slab unit: median {np.median(per):.2f} us  (cycles {np.median(cyc[sel] / np.ceil(nseg[sel] * q / 4.0)):.0f})")
print("longest workgroups: idx start kloop_end end cu")
for i in order[:12]:
    print(f"  {i:5d} {st[i]:7.2f} {kl[i]:7.2f} {en[i]:7.2f}  xcc {xcc[i]} se {se[i]} cu {cu[i]}  slabs {nseg[i]} quads {nq[i]} cycles {cyc[i]}")
late = np.argsort(-en)[:8]
print("last to finish:")
for i in late:
    print(f"  {i:5d} {st[i]:7.2f} {kl[i]:7.2f} {en[i]:7.2f}  xcc {xcc[i]} se {se[i]} cu {cu[i]}")
# per-CU busy: number of workgroups and span
import collections
per = collections.defaultdict(list)
for i in range(n):
    per[int(cuid[i])].append((st[i], en[i]))
cnt = np.array([len(v) for v in per.values()])
print("workgroups per CU: min %d mean %.2f max %d" % (cnt.min(), cnt.mean(), cnt.max()))
np.save(os.path.join(ROOT, "gpurun_out", "gemm_prof.npy"), out)
