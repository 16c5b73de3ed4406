"""ORACLE (test infrastructure only).

Known answers that do not depend on any tensor-network code:
  * dense Jordan-Wigner Hamiltonian for L <= 6 (checks the reduced MPO entry by entry),
  * sparse exact diagonalisation at fixed (N_up, N_dn) for L <= 12,
  * free-fermion closed form for U = 0 open chains,
  * exact Schmidt spectrum of the ED ground state resolved by (N_left, S).
Model = open-chain restatement of src/HubbardFunctions.jl:386-472 (see oracle/mpo.py).
"""
from __future__ import annotations

from itertools import combinations
from math import cos, pi

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from . import su2


# ----------------------------------------------------------------------------------------
def dense_hamiltonian(L: int, t, u, mu: float = 0.0):
    """4^L x 4^L matrix in the Jordan-Wigner product basis (site 1 most significant)."""
    lm = su2.local_matrices()
    I4, F = lm["id"], lm["F"]

    def site_op(mats):  # mats: dict site->4x4 (1-based)
        out = np.eye(1)
        for s in range(1, L + 1):
            out = np.kron(out, mats.get(s, I4))
        return out

    def c_op(i, spin):  # full annihilator c_{i spin}
        mats = {s: F for s in range(1, i)}
        mats[i] = lm["a_up"] if spin == 0 else lm["a_dn"]
        return site_op(mats)

    c = {(i, s): c_op(i, s) for i in range(1, L + 1) for s in (0, 1)}
    n = {i: site_op({i: lm["n"]}) for i in range(1, L + 1)}
    H = np.zeros((4 ** L, 4 ** L))
    for i in range(1, L + 1):
        H += u[0] * site_op({i: lm["docc"]}) - mu * n[i]
    for r in range(1, len(t) + 1):
        for i in range(1, L - r + 1):
            for s in (0, 1):
                h = c[(i, s)].T @ c[(i + r, s)]
                H += -t[r - 1] * (h + h.T)
    for r in range(1, len(u)):
        for i in range(1, L - r + 1):
            H += u[r] * n[i] @ n[i + r]
    return H


# ----------------------------------------------------------------------------------------
def free_fermion_energy(L: int, N_up: int, N_dn: int, t1: float = 1.0) -> float:
    """E0 of the open NN chain at U=0: fill the lowest levels -2 t cos(k pi/(L+1))."""
    eps = sorted(-2.0 * t1 * cos(k * pi / (L + 1)) for k in range(1, L + 1))
    return sum(eps[:N_up]) + sum(eps[:N_dn])


# ----------------------------------------------------------------------------------------
class SectorED:
    """Sparse ED in the (N_up, N_dn) sector.  Fermion order: all up modes (site 1..L) then all
    down modes; a basis state is (bits_up, bits_dn).  Energies are order independent."""

    def __init__(self, L, N_up, N_dn, t, u, mu=0.0):
        self.L, self.N_up, self.N_dn = L, N_up, N_dn
        self.up = [sum(1 << i for i in c) for c in combinations(range(L), N_up)]
        self.dn = [sum(1 << i for i in c) for c in combinations(range(L), N_dn)]
        self.iu = {b: n for n, b in enumerate(self.up)}
        self.idn = {b: n for n, b in enumerate(self.dn)}
        self.t, self.u, self.mu = list(t), list(u), mu

    def _hop_matrix(self, states, index):
        L = self.L
        rows, cols, vals = [], [], []
        for n, b in enumerate(states):
            for r in range(1, len(self.t) + 1):
                if self.t[r - 1] == 0.0:
                    continue
                for i in range(L - r):
                    j = i + r
                    for (src, dst) in ((i, j), (j, i)):
                        if (b >> src) & 1 and not (b >> dst) & 1:
                            lo, hi = min(src, dst), max(src, dst)
                            between = bin(b & (((1 << hi) - 1) ^ ((1 << (lo + 1)) - 1))).count("1")
                            nb = b ^ (1 << src) ^ (1 << dst)
                            rows.append(index[nb])
                            cols.append(n)
                            vals.append(-self.t[r - 1] * (-1) ** between)
        N = len(states)
        return sp.csr_matrix((vals, (rows, cols)), shape=(N, N))

    def hamiltonian(self):
        Hu = self._hop_matrix(self.up, self.iu)
        Hd = self._hop_matrix(self.dn, self.idn)
        nu, nd = len(self.up), len(self.dn)
        H = sp.kron(Hu, sp.identity(nd)) + sp.kron(sp.identity(nu), Hd)
        L = self.L
        occ_u = np.array([[(b >> i) & 1 for i in range(L)] for b in self.up], dtype=float)
        occ_d = np.array([[(b >> i) & 1 for i in range(L)] for b in self.dn], dtype=float)
        diag = self.u[0] * (occ_u @ occ_d.T)
        ntot = occ_u[:, None, :] + occ_d[None, :, :]            # [nu, nd, L]
        diag = diag - self.mu * ntot.sum(-1)
        for r in range(1, len(self.u)):
            if self.u[r] != 0.0:
                diag = diag + self.u[r] * np.einsum("abi,abi->ab", ntot[:, :, :L - r], ntot[:, :, r:])
        return (H + sp.diags(diag.reshape(-1))).tocsr()

    def ground_state(self):
        H = self.hamiltonian()
        if H.shape[0] <= 2000:
            w, v = np.linalg.eigh(H.toarray())
            return float(w[0]), v[:, 0]
        w, v = spla.eigsh(H, k=1, which="SA", tol=1e-13, ncv=40)
        return float(w[0]), v[:, 0]

    def schmidt_by_sector(self, psi, cut: int):
        """Singular values of psi across the bond after site `cut`, grouped by
        (N_left, N_left_up - N_left_dn).  Within a spin-S multiplet the same value recurs for
        every Sz, so the (N_left, 2S) spectrum is read off the Sz = S columns."""
        L = self.L
        mask = (1 << cut) - 1
        nu, nd = len(self.up), len(self.dn)
        psi = psi.reshape(nu, nd)
        groups = {}
        for a, bu in enumerate(self.up):
            for b, bd in enumerate(self.dn):
                lu, ld = bu & mask, bd & mask
                key = (bin(lu).count("1"), bin(ld).count("1"))
                g = groups.setdefault(key, ({}, {}, []))
                li = g[0].setdefault((lu, ld), len(g[0]))
                ri = g[1].setdefault((bu >> cut, bd >> cut), len(g[1]))
                g[2].append((li, ri, psi[a, b]))
        out = {}
        for (nlu, nld), (lmap, rmap, ent) in groups.items():
            M = np.zeros((len(lmap), len(rmap)))
            for li, ri, v in ent:
                M[li, ri] = v
            out[(nlu + nld, nlu - nld)] = np.linalg.svd(M, compute_uv=False)
        return out


def multiplet_spectrum(by_sector, tol=1e-12):
    """(N_left, twoS) -> descending Schmidt values of the spin multiplets, from the per
    (N_left, 2Sz) spectra: values at 2Sz = twoS minus those already present at twoS + 2."""
    out = {}
    Ns = sorted({k[0] for k in by_sector})
    for N in Ns:
        szs = sorted({k[1] for k in by_sector if k[0] == N and k[1] >= 0}, reverse=True)
        higher = np.array([])
        for sz in szs:
            vals = np.sort(by_sector[(N, sz)])[::-1]
            vals = vals[vals > tol]
            # remove (greedily) the values belonging to higher-spin multiplets
            rem = list(vals)
            for h in higher:
                j = int(np.argmin([abs(x - h) for x in rem]))
                rem.pop(j)
            if rem:
                out[(N, sz)] = np.array(sorted(rem, reverse=True))
            higher = vals
    return out
