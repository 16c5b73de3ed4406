"""ORACLE (test infrastructure only).

Finite-chain reduced MPO of the one-band Hubbard Hamiltonian, restating
`hamiltonian(::OB_Sim)` (src/HubbardFunctions.jl:386-472) for open boundaries:

    H = sum_i (u[0] n_up n_dn - mu n)_i                         (:424, :435)
      - sum_r t[r-1] sum_i sum_s (c+_{i s} c_{i+r s} + h.c.)    (:423, :437-440)
      + sum_{r>=1} u[r] sum_i n_i n_{i+r}                        (:441-444)

The reference sums over an infinite chain; here every term whose sites fit in 1..L is kept
(finite two-site DMRG for the L=... configs).

An MPO is a list (one per site) of dicts
    {"left": [(dN,k),...], "right": [(dN,k),...], "entries": [(wl, wr, opname, coef), ...]}
where (dN,k) label the particle number / doubled spin carried by a virtual level and
opname indexes oracle.su2.site_operators().  Level 0 of an interior bond is "nothing
applied yet", the last level is "term complete" (Jordan/upper-triangular form, App. A.3).
"""
from __future__ import annotations

from math import sqrt

import numpy as np

from . import su2

SQ2 = sqrt(2.0)


def hubbard_mpo(L: int, t, u, mu: float = 0.0):
    t = list(t)
    u = list(u)
    # channels: (kind, range r, step d) occupying the bond after `d` sites of a range-r term
    chans = []
    for r in range(1, len(t) + 1):
        if t[r - 1] != 0.0:
            for d in range(1, r + 1):
                chans.append(("hop+", r, d, (+1, 1)))   # c+ emitted on the left
            for d in range(1, r + 1):
                chans.append(("hop-", r, d, (-1, 1)))   # c emitted on the left
    for r in range(1, len(u)):
        if u[r] != 0.0:
            for d in range(1, r + 1):
                chans.append(("nn", r, d, (0, 0)))

    def bond_levels(b):
        """levels on the bond between site b and b+1 (1-based sites; b=0 and b=L are boundaries)"""
        if b == 0:
            return [("start",)]
        if b == L:
            return [("final",)]
        lv = [("start",)]
        for (kind, r, d, q) in chans:
            # the term started at site i = b - d + 1 and ends at i + r <= L
            i = b - d + 1
            if i >= 1 and i + r <= L:
                lv.append((kind, r, d))
        lv.append(("final",))
        return lv

    qn = {("start",): (0, 0), ("final",): (0, 0)}
    for (kind, r, d, q) in chans:
        qn[(kind, r, d)] = q

    mpo = []
    for site in range(1, L + 1):
        ll = bond_levels(site - 1)
        lr = bond_levels(site)
        il = {lv: n for n, lv in enumerate(ll)}
        ir = {lv: n for n, lv in enumerate(lr)}
        ent = []
        if ("start",) in il and ("start",) in ir:
            ent.append((il[("start",)], ir[("start",)], "id", 1.0))
        if ("final",) in il and ("final",) in ir:
            ent.append((il[("final",)], ir[("final",)], "id", 1.0))
        if ("start",) in il and ("final",) in ir:
            if u[0] != 0.0:
                ent.append((il[("start",)], ir[("final",)], "docc", u[0]))
            if mu != 0.0:
                ent.append((il[("start",)], ir[("final",)], "n", -mu))
        for lv in lr:
            if lv[0] in ("start", "final"):
                continue
            kind, r, d = lv
            if d == 1:      # term starts here
                op = {"hop+": "cdagF", "hop-": "Fc", "nn": "n"}[kind]
                ent.append((il[("start",)], ir[lv], op, 1.0))
            else:           # pass-through
                op = {"hop+": "F", "hop-": "F", "nn": "id"}[kind]
                ent.append((il[(kind, r, d - 1)], ir[lv], op, 1.0))
        for lv in ll:
            if lv[0] in ("start", "final"):
                continue
            kind, r, d = lv
            if d == r:      # term ends here
                if kind == "hop+":
                    ent.append((il[lv], ir[("final",)], "c", -t[r - 1] * SQ2))
                elif kind == "hop-":
                    ent.append((il[lv], ir[("final",)], "cdag", +t[r - 1] * SQ2))
                else:
                    ent.append((il[lv], ir[("final",)], "n", u[r]))
        mpo.append({"left": [qn[lv] for lv in ll], "right": [qn[lv] for lv in lr], "entries": ent})
    return mpo


def mpo_to_dense(mpo):
    """Expand the reduced MPO with explicit CG tensors and contract it to the 4^L x 4^L matrix
    (Jordan-Wigner product basis, site 1 most significant).  Checks only; L <= 6."""
    ops = su2.site_operators()
    cur = None  # [levelsfull, row, col]
    for site, W in enumerate(mpo):
        offl = np.cumsum([0] + [k + 1 for (_, k) in W["left"]])
        offr = np.cumsum([0] + [k + 1 for (_, k) in W["right"]])
        Wf = np.zeros((offl[-1], 4, 4, offr[-1]))
        for (wl, wr, name, coef) in W["entries"]:
            kop = ops[name][0]
            comps = su2.expand_site_operator(name)
            kl = W["left"][wl][1]
            kr = W["right"][wr][1]
            C = su2.cg_tensor(kl, kop, kr)   # [ml, q, mr]
            for q in range(kop + 1):
                Wf[offl[wl]:offl[wl + 1], :, :, offr[wr]:offr[wr + 1]] += coef * np.einsum(
                    "lr,ps->lpsr", C[:, q, :], comps[q])
        if cur is None:
            assert Wf.shape[0] == 1
            cur = Wf[0].transpose(2, 0, 1)            # [r, p, s]
        else:
            cur = np.einsum("lab,lpsr->rapbs", cur, Wf)
            n = cur.shape[1] * cur.shape[2]
            cur = cur.reshape(cur.shape[0], n, n)
    assert cur.shape[0] == 1
    return cur[0]
