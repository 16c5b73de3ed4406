"""Initial matrix-product states for the finite-chain engine.

`random_mps` mirrors `initialize_mps` (src/HubbardFunctions.jl:917-959): virtual spaces are the
intersection of what can be fused from the left and from the right, capped PER SECTOR at
`max_dimension` multiplets (src:931-938), tensors random; here for an open chain with a target
total sector and returned in right-canonical form (Euclidean "tilde" normalisation).
Both symmetry modes of the fixed-filling models: SU(2) x U(1) (default) and U(1) x U(1) (`spin=true`).
"""
from __future__ import annotations

import numpy as np

from .models import SU2U1
from .sectors import full_bonds


def _right_canonical_site(rng, bl, br, sym, shrink=True):
    """random site tensor between bond tables bl, br (dicts, bl may shrink): rows of every (c ; (s, b)) matrix orthonormal"""
    T = {}
    for c in sorted(bl):
        cols = [(s, b) for s in range(sym.n_site) for b in sym.fuse(c, s) if b in br]
        ncols = sum(br[b] for (_, b) in cols)
        if ncols == 0:
            if shrink:
                del bl[c]
            continue
        nc = bl[c]
        if shrink:
            nc = min(nc, ncols)
            bl[c] = nc
        M = rng.standard_normal((nc, ncols)) + 1j * rng.standard_normal((nc, ncols))
        if shrink:                                  # right-canonical: orthonormal rows
            q, _ = np.linalg.qr(M.conj().T)
            M = q.conj().T
        off = 0
        for (s, b) in cols:
            T[(c, s, b)] = np.ascontiguousarray(M[:, off:off + br[b]])
            off += br[b]
    return T


def random_mps(nsites, target, max_dimension, seed=1234, max_twoS=6, sym=SU2U1):
    """returns (bonds: list[dict], tensors: list[dict (l,s,r) -> ndarray]).
    max_twoS = 6 is the reference's S <= 3 cap on the initial virtual spaces (src:933; |2 Sz| <= 6 in the spinful mode)."""
    rng = np.random.default_rng(seed)
    bonds = [{sec: min(n, max_dimension) for sec, n in b.dims.items() if abs(sec[1]) <= max_twoS}
             for b in full_bonds(nsites, target, sym)]
    tensors = [None] * nsites
    for i in range(nsites - 1, -1, -1):
        tensors[i] = _right_canonical_site(rng, bonds[i], bonds[i + 1], sym)
    for i in range(nsites):
        tensors[i] = {k: v for k, v in tensors[i].items() if k[0] in bonds[i] and k[2] in bonds[i + 1]}
    return bonds, tensors


def random_window(nsites, bond_left, bond_right, max_dimension, seed=1234, max_twoS=8, sym=SU2U1):
    """random state of `nsites` sites between two FIXED boundary bond tables (the bases of the left and right
    environment blocks of an iDMRG window, hubbardtn_amd/idmrg.py).  Internal virtual spaces = sectors reachable
    from the left table and co-reachable from the right one, capped per sector; sites 1.. are right-canonical,
    site 0 carries the (unnormalised) centre.  Same return convention as random_mps."""
    rng = np.random.default_rng(seed)
    fwd = [dict(bond_left)]
    for _ in range(nsites):
        nxt = {}
        for c, n in fwd[-1].items():
            for s in range(sym.n_site):
                for b in sym.fuse(c, s):
                    nxt[b] = min(nxt.get(b, 0) + n, 1 << 30)
        fwd.append(nxt)
    bwd = [None] * (nsites + 1)
    bwd[nsites] = dict(bond_right)
    for i in range(nsites - 1, -1, -1):
        prv = {}
        for b, n in bwd[i + 1].items():
            for s in range(sym.n_site):
                for c in sym.split(b, s):
                    prv[c] = min(prv.get(c, 0) + n, 1 << 30)
        bwd[i] = prv
    bonds = []
    for i in range(nsites + 1):
        if i == 0:
            bonds.append(dict(bond_left))
        elif i == nsites:
            bonds.append(dict(bond_right))
        else:
            bonds.append({c: min(fwd[i][c], bwd[i][c], max_dimension) for c in fwd[i]
                          if c in bwd[i] and abs(c[1]) <= max_twoS})
    tensors = [None] * nsites
    for i in range(nsites - 1, -1, -1):
        tensors[i] = _right_canonical_site(rng, bonds[i], bonds[i + 1], sym, shrink=i > 0)
    for i in range(nsites):
        tensors[i] = {k: v for k, v in tensors[i].items() if k[0] in bonds[i] and k[2] in bonds[i + 1]}
    return bonds, tensors
