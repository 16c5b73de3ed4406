"""Thin Python face of the C++ sweep engine (hubbardtn_amd/csrc/htn_engine.cpp behind include/hubbardtn_hip.h).

Everything below `find_groundstate(psi0, H, alg)` (src/HubbardFunctions.jl:1010) -- sector layouts, recoupling
coefficients, task lists, theta formation, Lanczos, per-sector SVD, the global truncation rule, write-back,
environment transfer, the sweep loop -- runs inside the library (`htn_mps_create`, `htn_bond_update`,
`htn_dmrg2_sweep`).  This module only converts the host-side model (list of models.MPOSite) and the initial state
(bond tables + block dictionaries) into the ABI's tables and reads results back.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import abi
from .models import SU2U1
from .sectors import Bond


@dataclass
class BondStats:
    bond: int
    direction: int
    energy: float
    n_matvec: int
    residual: float
    trunc_weight: float
    chi_full: int
    multiplets: int
    theta_size: int
    apply_flops: int
    apply_bytes: int
    svd_flops: int
    jacobi_sweeps: int
    n_tiles: int
    n_segs: int
    t_plan: float = 0.0
    t_total: float = 0.0
    t_lanczos: float = 0.0
    t_svd: float = 0.0
    t_env: float = 0.0
    matvec_ms: float = 0.0


def _stats(rec) -> BondStats:
    return BondStats(**{k: (float(rec[k]) if rec.dtype[k].kind == "f" else int(rec[k])) for k in rec.dtype.names})


class CMpo:
    """htn_mpo handle built from a models.MPO (list of MPOSite + symmetry; a plain list means SU(2) x U(1) x fZ2)"""

    def __init__(self, ops, sites):
        self.ops, self.lib = ops, ops.lib
        self.sym = msym = getattr(sites, "sym", SU2U1)
        SITE_MULT, SITE_OPS = msym.site_mult, msym.site_ops
        names = sorted({e[2] for s in sites for e in s.entries})
        self.op_index = {n: k for k, n in enumerate(names)}
        optab = np.zeros(max(len(names), 1), dtype=abi.SITE_OP_DT)
        for n, k in self.op_index.items():
            kk, dN, red = SITE_OPS[n]
            optab[k]["k"], optab[k]["dN"] = kk, dN
            r = np.zeros((abi.MAX_SITE, abi.MAX_SITE))
            r[:red.shape[0], :red.shape[1]] = red
            optab[k]["red"] = r.reshape(-1)
        n = len(sites)
        levels, level_ptr = [], [0]
        bonds = [sites[0].left] + [s.right for s in sites]
        for i in range(1, n):
            if list(sites[i].left) != list(sites[i - 1].right):
                raise ValueError(f"MPO bond {i}: left levels of site {i} differ from right levels of site {i - 1}")
        for lv in bonds:
            levels.extend(lv)
            level_ptr.append(level_ptr[-1] + len(lv))
        ent = np.zeros(max(sum(len(s.entries) for s in sites), 1), dtype=abi.MPO_ENTRY_DT)
        entry_ptr, q = [0], 0
        for s in sites:
            for (wl, wr, name, coef) in s.entries:
                c = complex(coef)
                ent[q] = (wl, wr, self.op_index[name], 0, c.real, c.imag)
                q += 1
            entry_ptr.append(q)
        sym = abi.Symmetry()
        sym.kind, sym.n_site = msym.kind, len(SITE_MULT)
        for s, (N, j) in enumerate(SITE_MULT):
            sym.site_N[s], sym.site_j[s] = N, j
        lv = np.ascontiguousarray(np.array(levels, dtype=np.int32).reshape(-1, 2))
        lp = np.array(level_ptr, dtype=np.int32)
        ep = np.array(entry_ptr, dtype=np.int32)
        h = C.c_void_p()
        abi.check(self.lib, self.lib.htn_mpo_create(ops.ctx, C.byref(sym), n, optab.ctypes.data, len(names), lp.ctypes.data,
                                                    lv.ctypes.data, ep.ctypes.data, ent.ctypes.data, C.byref(h)),
                  "htn_mpo_create")
        self.handle = h
        self.sites = sites

    def __del__(self):
        h, self.handle = getattr(self, "handle", None), None
        if h:
            self.lib.htn_mpo_destroy(h)


class DMRG2:
    """finite two-site DMRG on reduced SU(2) x U(1) tensors -- handle of an `htn_mps` inside the library.

    ops        : context provider (hubbardtn_amd.device.HipOps in the product; `.lib`, `.ctx`)
    mpo        : list[models.MPOSite]
    bonds      : list of {sector: count} for bonds 0..L (initial state)
    tensors    : list of {(l, s, r): ndarray[n_l, n_r]} right-canonical initial site tensors
    chi_full   : truncdim(D) in TensorKit's `dim` units (sum (2S+1) n), or None
    cutoff     : truncbelow(eta) Schmidt-value cut (10^-svalue, src:1007), or 0
    left_env / right_env : boundary environments (flat arrays in the library's block order, `env_data`) of an iDMRG
                 window; None = open end
    Sharding (sector-parallel apply) is a property of the context: HipOps.set_comm / set_exchange.
    """

    def __init__(self, ops, mpo, bonds, tensors, chi_full=None, cutoff=0.0, krylovdim=30, lanczos_tol=1e-12,
                 maxrestart=3, weighting="sqrtdim", jacobi_tol=1e-14, jacobi_max_sweeps=40, left_env=None, right_env=None):
        self.ops, self.lib = ops, ops.lib
        self.cmpo = mpo if isinstance(mpo, CMpo) else CMpo(ops, mpo)
        self.mpo = self.cmpo.sites
        self.sym = self.cmpo.sym
        self.L = len(self.mpo)
        self.chi_full, self.cutoff, self.weighting = chi_full, cutoff, weighting
        self.krylovdim, self.lanczos_tol, self.maxrestart = krylovdim, lanczos_tol, maxrestart
        self.jacobi_tol, self.jacobi_max_sweeps = jacobi_tol, jacobi_max_sweeps
        self.rank_cut = 0.0          # optional rank-revealing QR cut (OFF: the 1e-8 spectra parity needs it off), DESIGN.md 4
        self.svd_split = 0           # tests: force small blocks through the large-block SVD path
        self.profile = False
        self.energy = None
        self.stats = []
        L = self.L
        bond_ptr, secs = [0], []
        for b in bonds:
            items = sorted((k, int(v)) for k, v in dict(b).items() if v > 0)
            secs.extend((N, j, n) for (N, j), n in items)
            bond_ptr.append(len(secs))
        sec_arr = np.array(secs, dtype=np.int32).reshape(-1, 3).view(abi.SECTOR_DT).reshape(-1)
        subs, sub_ptr, data_ptr, chunks, pos = [], [0], [0], [], 0
        for i in range(L):
            off = 0
            for (l, s, r), blk in tensors[i].items():
                blk = np.asarray(blk, dtype=np.complex128)
                m, n = blk.shape
                subs.append((l[0], l[1], s, r[0], r[1], m, off))
                chunks.append(np.asfortranarray(blk).reshape(-1, order="F"))
                off += m * n
            sub_ptr.append(len(subs))
            pos += off
            data_ptr.append(pos)
        sub_arr = np.array(subs, dtype=np.int64).reshape(-1, 7)
        sb = np.zeros(max(len(subs), 1), dtype=abi.SUBBLOCK_DT)
        if len(subs):
            for k, f in enumerate(("lN", "lj", "s", "rN", "rj", "ld", "off")):
                sb[f] = sub_arr[:, k]
        data = np.concatenate(chunks) if chunks else np.zeros(1, dtype=np.complex128)
        bp = np.array(bond_ptr, dtype=np.int32)
        sp = np.array(sub_ptr, dtype=np.int32)
        dp = np.array(data_ptr, dtype=np.int64)
        le = None if left_env is None else np.ascontiguousarray(left_env, dtype=np.complex128)
        re_ = None if right_env is None else np.ascontiguousarray(right_env, dtype=np.complex128)
        h = C.c_void_p()
        abi.check(self.lib, self.lib.htn_mps_create(ops.ctx, self.cmpo.handle, L, bp.ctypes.data, sec_arr.ctypes.data,
                                                    sp.ctypes.data, sb.ctypes.data, dp.ctypes.data, data.ctypes.data,
                                                    None if le is None else le.ctypes.data,
                                                    None if re_ is None else re_.ctypes.data, C.byref(h)),
                  "htn_mps_create")
        self.handle = h

    def __del__(self):
        h, self.handle = getattr(self, "handle", None), None
        if h:
            self.lib.htn_mps_destroy(h)

    def _check(self, rc, what):
        """status of a library call that may have run the context's exchange hook: an exception raised inside the hook
        (stored by the context provider, the C side only saw "failed") is re-raised here in preference to the generic
        library error"""
        chk = getattr(self.ops, "check_exchange", None)
        if chk is not None:
            chk()
        abi.check(self.lib, rc, what)

    # ---- options ----------------------------------------------------------------------------------
    def _opts(self, cutoff=None):
        o = abi.SweepOpts()
        o.chi_full = int(self.chi_full) if self.chi_full else 0
        o.weighting = 0 if self.weighting == "sqrtdim" else 1
        o.cutoff = float(self.cutoff if cutoff is None else cutoff)
        o.krylovdim, o.maxrestart, o.lanczos_tol = int(self.krylovdim), int(self.maxrestart), float(self.lanczos_tol)
        o.jacobi_tol, o.jacobi_max_sweeps = float(self.jacobi_tol), int(self.jacobi_max_sweeps)
        o.svd_split_elems, o.rank_cut, o.profile = int(self.svd_split), float(self.rank_cut), 1 if self.profile else 0
        return o

    # ---- state queries ----------------------------------------------------------------------------
    @property
    def bonds(self):
        return [self.bond(b) for b in range(self.L + 1)]

    def bond(self, b) -> Bond:
        n = self.lib.htn_mps_bond(self.handle, b, None)
        arr = np.zeros(max(n, 1), dtype=abi.SECTOR_DT)
        self.lib.htn_mps_bond(self.handle, b, arr.ctypes.data)
        return Bond({(int(r["N"]), int(r["j"])): int(r["count"]) for r in arr[:n]}, self.sym)

    def bond_dims(self):
        """`dim_state` analogue (src/HubbardFunctions.jl:1399-1405): TensorKit dim of each bond"""
        return [b.dim_full for b in self.bonds]

    @property
    def spectra(self):
        return {b: s for b in range(1, self.L) for s in [self.spectrum(b)] if s}

    def spectrum(self, b):
        """Schmidt values of the last update of bond b: {sector: descending array}"""
        n = self.lib.htn_mps_spectrum(self.handle, b, None, None)
        if n <= 0:
            return {}
        secs = np.zeros(n, dtype=abi.SECTOR_DT)
        vals = np.zeros(n)
        self.lib.htn_mps_spectrum(self.handle, b, secs.ctypes.data, vals.ctypes.data)
        out, pos = {}, 0
        for r in secs:
            if pos >= n:
                break
            c = int(r["count"])
            out[(int(r["N"]), int(r["j"]))] = vals[pos:pos + c].copy()
            pos += c
        return out

    def download_site(self, i):
        """{(l, s, r): ndarray[n_l, n_r]} of site i as stored (left or right layout)"""
        nb = self.lib.htn_mps_get_site(self.handle, i, None, None)
        size = self.lib.htn_mps_site_size(self.handle, i, None)
        subs = np.zeros(max(nb, 1), dtype=abi.SUBBLOCK_DT)
        flat = np.zeros(max(size, 1), dtype=np.complex128)
        if self.lib.htn_mps_get_site(self.handle, i, subs.ctypes.data, flat.ctypes.data) < 0:
            raise abi.HtnError(self.lib.htn_last_error().decode())
        out = {}
        for sb in subs[:nb]:
            l, r = (int(sb["lN"]), int(sb["lj"])), (int(sb["rN"]), int(sb["rj"]))
            m, n = self.bond_of_site(i, 0)[l], self.bond_of_site(i, 1)[r]
            idx = int(sb["off"]) + np.arange(m)[:, None] + int(sb["ld"]) * np.arange(n)[None, :]
            out[(l, int(sb["s"]), r)] = flat[idx].copy()
        return out

    def bond_of_site(self, i, side):
        key = ("_b", i + side)
        cache = self.__dict__.setdefault("_bond_cache", {})
        # bonds change with every update: the cache is dropped in update_bond / sweep
        if key not in cache:
            cache[key] = self.bond(i + side)
        return cache[key]

    def site_kind(self, i):
        k = C.c_int32(0)
        self.lib.htn_mps_site_size(self.handle, i, C.byref(k))
        return chr(k.value)

    def env_data(self, side, b):
        """flat environment (library block order) on bond b; side 'L' / 'R'"""
        sd = 0 if side == "L" else 1
        n = self.lib.htn_mps_env_size(self.handle, sd, b)
        if n < 0:
            raise abi.HtnError(f"environment {side} of bond {b} does not exist")
        flat = np.zeros(max(n, 1), dtype=np.complex128)
        abi.check(self.lib, self.lib.htn_mps_get_env(self.handle, sd, b, flat.ctypes.data), "htn_mps_get_env")
        return flat[:n]

    def env_bond(self, side, b) -> Bond:
        """the bond table the environment on bond b was built on (left: last rightward update of that bond, right: last
        leftward update -- a truncation by dimension may keep different counts in between)"""
        sd = 0 if side == "L" else 1
        n = self.lib.htn_mps_env_bond(self.handle, sd, b, None)
        if n < 0:
            raise abi.HtnError(f"environment {side} of bond {b} does not exist")
        arr = np.zeros(max(n, 1), dtype=abi.SECTOR_DT)
        self.lib.htn_mps_env_bond(self.handle, sd, b, arr.ctypes.data)
        return Bond({(int(r["N"]), int(r["j"])): int(r["count"]) for r in arr[:n]}, self.sym)

    def download_env(self, side, b):
        """{(x, w, y): matrix}: left env keys (bra, w, ket) -> [n_bra, n_ket]; right env (ket, w, bra) -> [n_ket, n_bra]"""
        sd = 0 if side == "L" else 1
        nb = self.lib.htn_mps_env_blocks(self.handle, sd, b, None)
        blk = np.zeros(max(nb, 1), dtype=abi.ENV_BLOCK_DT)
        self.lib.htn_mps_env_blocks(self.handle, sd, b, blk.ctypes.data)
        flat = self.env_data(side, b)
        out = {}
        for r in blk[:nb]:
            m, n, off = int(r["rows"]), int(r["cols"]), int(r["off"])
            out[((int(r["aN"]), int(r["aj"])), int(r["w"]), (int(r["bN"]), int(r["bj"])))] = \
                flat[off:off + m * n].reshape(n, m).T.copy()
        return out

    def theta(self, i):
        n = self.lib.htn_mps_theta_size(self.handle, i)
        out = np.zeros(n, dtype=np.complex128)
        abi.check(self.lib, self.lib.htn_mps_get_theta(self.handle, i, out.ctypes.data), "htn_mps_get_theta")
        return out

    def apply_heff(self, i, x):
        """y = H_eff(bond i, i+1) x on host vectors in the library's theta layout"""
        x = np.ascontiguousarray(x, dtype=np.complex128)
        n = self.lib.htn_mps_theta_size(self.handle, i)
        assert x.shape == (n,)
        y = np.zeros(n, dtype=np.complex128)
        self._check(self.lib.htn_heff2_apply(self.handle, i, x.ctypes.data, y.ctypes.data), "htn_heff2_apply")
        return y

    def plan_apply_dump(self, i, stage):
        """(tiles, segs, z_size, flops) of the H_eff apply on bond i as the kernels receive them (tests)"""
        nt, nsg, zs, fl = C.c_int32(0), C.c_int32(0), C.c_int64(0), C.c_int64(0)
        abi.check(self.lib, self.lib.htn_plan_apply_dump(self.handle, i, stage, C.byref(nt), None, C.byref(nsg), None,
                                                         C.byref(zs), C.byref(fl)), "htn_plan_apply_dump")
        tiles = np.zeros(max(nt.value, 1), dtype=abi.TILE_DT)
        segs = np.zeros(max(nsg.value, 1), dtype=abi.SEG_DT)
        abi.check(self.lib, self.lib.htn_plan_apply_dump(self.handle, i, stage, C.byref(nt), tiles.ctypes.data,
                                                         C.byref(nsg), segs.ctypes.data, C.byref(zs), C.byref(fl)),
                  "htn_plan_apply_dump")
        return tiles[:nt.value], segs[:nsg.value], zs.value, fl.value

    @property
    def cache_hits(self):
        return self._cache_stats()[0]

    @property
    def cache_misses(self):
        return self._cache_stats()[1]

    def _cache_stats(self):
        h, m = C.c_int64(0), C.c_int64(0)
        self.lib.htn_mps_cache_stats(self.handle, C.byref(h), C.byref(m))
        return h.value, m.value

    # ---- updates ----------------------------------------------------------------------------------
    def update_bond(self, i, direction, placement, optimise=True, record=True, cutoff=None):
        """optimise sites (i, i+1); placement 'right': A_i = U, centre S V^H on i+1 (+ left env); 'left': centre U S on
        i, B_{i+1} = V^H (+ right env).  optimise=False only moves the centre (E = <theta|H|theta>)."""
        self.__dict__.pop("_bond_cache", None)
        st = np.zeros(1, dtype=abi.BOND_STATS_DT)
        o = self._opts(cutoff)
        self._check(self.lib.htn_bond_update(self.handle, i, direction, 0 if placement == "right" else 1,
                                             1 if optimise else 0, C.byref(o), st.ctypes.data), "htn_bond_update")
        s = _stats(st[0])
        if record:
            self.stats.append(s)
            self.energy = s.energy
        return s.energy

    def sweep(self):
        """one sweep in MPSKit's DMRG2 order (bonds 0..L-2 rightwards, L-3..0 leftwards), one library call"""
        self.__dict__.pop("_bond_cache", None)
        n = 2 * self.L - 3
        st = np.zeros(n, dtype=abi.BOND_STATS_DT)
        E = C.c_double(0.0)
        o = self._opts()
        self._check(self.lib.htn_dmrg2_sweep(self.handle, C.byref(o), st.ctypes.data, C.byref(E)), "htn_dmrg2_sweep")
        self.stats.extend(_stats(r) for r in st)
        self.energy = E.value
        return self.energy

    def site_probabilities(self):
        """-> P[L, n_site]: probability of every site multiplet on every site.  Call after sweep() (centre on site 0,
        sites >= 1 right-canonical).  The centre is carried through the chain without optimisation; with the centre on
        site i the probability of site multiplet s is the squared norm of the (., s, .) blocks (tilde normalisation).
        Leaves stats/energy alone."""
        L = self.L
        P = np.zeros((L, self.sym.n_site))

        def read(i):
            p = np.zeros(self.sym.n_site)
            for (l, s, r), blk in self.download_site(i).items():
                p[s] += float(np.sum(np.abs(blk) ** 2))
            P[i] = p / p.sum()
        read(0)
        for i in range(L - 1):          # moving the centre must not truncate by value: cutoff 0
            self.update_bond(i, +1, "right", optimise=False, record=False, cutoff=0.0)
            read(i + 1)
        for i in range(L - 2, -1, -1):  # back to the post-sweep convention
            self.update_bond(i, -1, "left", optimise=False, record=False, cutoff=0.0)
        return P

    def site_occupations(self):
        """-> (n, d): <n_i> and the double occupancy <n_up n_dn>_i of every site (density_state, src:1495-1523)"""
        P = self.site_probabilities()
        Ns = np.array(self.sym.site_electrons, dtype=float)                     # electrons of every site multiplet
        return P @ Ns, P[:, -1].copy()                                            # (the last multiplet is the doubly occupied one)

    def spin_occupations(self):
        """-> (n_up, n_dn) per site; spinful U(1) x U(1) mode only (density_spin, src:1412-1456)"""
        if self.sym.su2:
            raise ValueError("This system is spin independent.")                # the reference's error text (src:1424)
        P = self.site_probabilities()
        return P[:, 1] + P[:, 3], P[:, 2] + P[:, 3]

    def bond_energies(self):
        """<psi| H |psi> of the state AS STORED (truncated), evaluated by a non-optimising pass: returns (E_total,
        per-bond running values).  Call after sweep() (centre on site 0).  The expectation value is the same number at
        every bond (the centre move is exact when nothing is truncated: cutoff 0, and the dimension limit is not
        reached by a state that already obeys it); the list documents that."""
        vals = []
        for i in range(self.L - 1):
            vals.append(self.update_bond(i, +1, "right" if i < self.L - 2 else "left", optimise=False, record=False,
                                         cutoff=0.0))
        for i in range(self.L - 3, -1, -1):
            vals.append(self.update_bond(i, -1, "left", optimise=False, record=False, cutoff=0.0))
        return vals[-1], vals

    def svd_cut(self, chi_full):
        """truncate every bond to truncdim(chi_full) by SVD alone (MPSKit `changebonds(psi, SvdCut(trscheme))`,
        used at src:1363-1365): one pass of centre moves without optimisation at the new limit, then a second,
        non-truncating pass that evaluates <psi|H|psi> of the TRUNCATED state, which is returned.  Call after sweep()."""
        self.chi_full = int(chi_full)
        for i in range(self.L - 1):
            self.update_bond(i, +1, "right" if i < self.L - 2 else "left", optimise=False, record=False, cutoff=0.0)
        for i in range(self.L - 3, -1, -1):
            self.update_bond(i, -1, "left", optimise=False, record=False, cutoff=0.0)
        E, _ = self.bond_energies()
        self.energy = E
        return E

    def site_energies(self):
        """genuine <psi|H|psi> of the state as stored, split per site: e_i = energy of all Hamiltonian terms that END
        on site i (on-site terms of i, and every two-site term whose right-most site is i); sum(e) = <psi|H|psi>.
        Evaluated by one non-optimising rightward pass: with the centre in Schmidt form on bond b, the completed-term
        level of the left environment gives E(sites < b) = sum_c sum_k s~_{c,k}^2 GL_final[c][k, k]; the last site closes
        with the total <theta|H|theta>.  Call after sweep() (centre on site 0); the state is left as found."""
        L = self.L
        run = np.zeros(L)                                   # run[i] = energy of the terms inside sites 0..i
        E_tot = None
        nfin = len(self.mpo[0].right) - 1
        for i in range(L - 1):
            E_tot = self.update_bond(i, +1, "right", optimise=False, record=False, cutoff=0.0)
            spec = self.spectrum(i + 1)
            env = self.download_env("L", i + 1)
            wfin = len(self.mpo[i].right) - 1               # 'term complete' level of the MPO bond right of site i
            acc = 0.0
            for (bra, w, ket), M in env.items():
                if w == wfin and bra == ket and bra in spec:
                    s2 = self.sym.qdim(bra) * np.asarray(spec[bra]) ** 2
                    acc += float(np.real(np.sum(np.diag(M)[:len(s2)] * s2)))
            run[i] = acc
        run[L - 1] = E_tot
        for i in range(L - 2, -1, -1):                      # back to the post-sweep convention (centre on site 0)
            self.update_bond(i, -1, "left", optimise=False, record=False, cutoff=0.0)
        e = np.diff(np.concatenate([[0.0], run]))
        return e
