"""how many singular values of the large two-site blocks are numerically negligible at chi = 1024?"""
import sys
import numpy as np
sys.path.insert(0, ".")
from hubbardtn_amd import engine, models, mps
from hubbardtn_amd.device import HipOps

ops = HipOps(0)
L = 64
mpo = models.hamiltonian(models.OB_Sim([1.0], [4.0]), L)
bonds, tens = mps.random_mps(L, (L, 0), 4)
eng = engine.DMRG2(ops, mpo, bonds, tens, chi_full=16, lanczos_tol=1e-6)
for chi, n in [(16, 8), (32, 4), (64, 4), (128, 2), (256, 2), (512, 2)]:
    eng.chi_full = chi
    for _ in range(n):
        eng.sweep()
eng.chi_full, eng.lanczos_tol = 1024, 1e-10
eng.sweep()
eng.sweep()
rec = []
orig = ops.jacobi_svd


def spy(G, Vj, S, d_desc, nb, max_m, ms, tol, info, desc_host=None):
    orig(G, Vj, S, d_desc, nb, max_m, ms, tol, info, desc_host=desc_host)
    s = ops.to_host(S)
    for k in range(nb):
        so, n = int(desc_host[k]["s_off"]), int(desc_host[k]["n"])
        if n >= 100:
            v = np.sort(s[so:so + n])[::-1]
            rec.append((n, [int((v > v[0] * t).sum()) for t in (1e-8, 1e-10, 1e-12, 1e-14)], v[0]))


ops.jacobi_svd = spy
eng.sweep()
ns = np.array([r[0] for r in rec])
above = np.array([r[1] for r in rec])
print("large blocks (n >= 100):", len(rec), " mean n", ns.mean())
for j, t in enumerate((1e-8, 1e-10, 1e-12, 1e-14)):
    print(f"  values > {t:g} * block max: mean fraction {np.mean(above[:, j] / ns):.3f}   sum n'^2 / sum n^2 = {np.sum(above[:, j] ** 2) / np.sum(ns ** 2):.3f}")
