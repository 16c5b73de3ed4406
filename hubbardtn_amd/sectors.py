"""Sector bookkeeping shared by the host-side state initialisers and drivers: fusion rules of the site multiplets and
ordered bond tables.  (The contraction planner itself lives in the C++ library, hubbardtn_amd/csrc/htn_plan.cpp.)

Sector = (N, j): particle number (fermion parity = N mod 2; N replaces the reference's shifted charge
k = N Q - P sites, src/HubbardFunctions.jl:251) and j = doubled spin 2S (SU(2) x U(1), default) or 2 Sz (`spin=true`,
U(1) x U(1)).  All functions take the symmetry descriptor (models.Symmetry); the default is SU(2) x U(1).
"""
from __future__ import annotations

from .models import SU2U1


def fuse(sec, s, sym=SU2U1):
    """sectors reachable from `sec` by adding site multiplet s"""
    return sym.fuse(sec, s)


def split(sec, s, sym=SU2U1):
    """sectors c with c (x) s -> sec"""
    return sym.split(sec, s)


class Bond:
    """ordered sector table of one virtual bond: sector (N, j) -> multiplet count"""

    def __init__(self, dims: dict, sym=SU2U1):
        items = sorted((k, int(v)) for k, v in dims.items() if v > 0)
        self.secs = [k for k, _ in items]
        self.dims = {k: v for k, v in items}
        self._key = tuple(items)
        self.sym = sym

    def __contains__(self, sec):
        return sec in self.dims

    def __getitem__(self, sec):
        return self.dims[sec]

    def __iter__(self):
        return iter(self.secs)

    def __eq__(self, other):
        return isinstance(other, Bond) and self.dims == other.dims

    def key(self):
        return self._key

    @property
    def dim_full(self):
        """TensorKit `dim` (SU(2)-expanded), the unit `dim_state` prints (src:1399-1405)"""
        return sum(self.sym.qdim(sec) * n for sec, n in self.dims.items())

    @property
    def multiplets(self):
        return sum(self.dims.values())


def full_bonds(nsites, target, sym=SU2U1):
    """exact (untruncated) bond tables of an open chain with total sector `target`"""
    left = [{(0, 0): 1}]
    for _ in range(nsites):
        nxt = {}
        for sec, n in left[-1].items():
            for s in range(sym.n_site):
                for c in sym.fuse(sec, s):
                    nxt[c] = nxt.get(c, 0) + n
        left.append(nxt)
    right = [{target: 1}]
    for _ in range(nsites):
        prv = {}
        for sec, n in right[-1].items():
            for s in range(sym.n_site):
                for c in sym.split(sec, s):
                    prv[c] = prv.get(c, 0) + n
        right.append(prv)
    right = right[::-1]
    return [Bond({sec: min(left[i][sec], right[i][sec]) for sec in left[i] if sec in right[i]}, sym)
            for i in range(nsites + 1)]
