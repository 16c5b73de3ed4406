"""TEST STATEMENT of the contraction planner (the product's planner is C++: hubbardtn_amd/csrc/htn_plan.cpp;
tests/test_cplan_cpu.py asserts that both emit byte-identical task lists).

Host planner: turns sector tables + a reduced MPO into the task lists the HIP kernels execute.

This is the MI355X-first replacement for what TensorKit does with fusion trees and tree
transformers around every contraction: instead of permuting
tensors between matricisations at run time, every contraction on the two-site DMRG path is
compiled ONCE per bond geometry into a list of (output tile, segment) records for the grouped
GEMM kernel; recoupling coefficients (closed-form 9j, hubbardtn_amd/wigner.py) become the
segments' alpha.

Layouts (all column-major, complex128, Euclidean "tilde" normalisation -- DESIGN.md):
  two-site tensor  : per mid sector c one dense matrix M_c[(a,s1) rows ; (s2,b) cols]
  left site tensor : per right sector c  matrix [(a,s) rows ; n_c]           (= U of the SVD)
  right site tensor: per left sector c   matrix [n_c ; (s,b) cols]           (= V^H of the SVD)
  left env   L     : blocks (a', w, a)  -> [n_a', n_a]; level 0 ('start') implicit identity
  right env  Rt    : blocks (b, w, b')  -> [n_b, n_b'] (stored transposed); last level implicit identity
"""
from __future__ import annotations

from dataclasses import dataclass, field
from functools import lru_cache
from math import sqrt

import numpy as np

from hubbardtn_amd.abi import COPY_DT, OP_C, OP_N, OP_T, SEG_COPY, SEG_DT, SEG_GEMM, SVD_DT, TILE_DT, HTN_TILE
from hubbardtn_amd.models import SITE_MULT, SITE_OPS
from hubbardtn_amd.sectors import Bond, full_bonds, fuse, split
from ref_wigner import triangle, wigner9j

# buffer-table slots shared by all plans
BUF_X, BUF_Y, BUF_L, BUF_R, BUF_Z, BUF_S1, BUF_S2, BUF_AUX = range(8)


# ----------------------------------------------------------------------------------------------
# recoupling coefficients (validated against oracle/su2.py in tests/test_wigner.py)
# ----------------------------------------------------------------------------------------------
@lru_cache(maxsize=None)
def coef_left(jbp, k, jb, jsp, js, kop, jap, kp, ja):
    """L'[a',w',a] += coef * A[b',s',a']^+ L[b',w,b] W[w,s',s,w'] A[b,s,a]"""
    return sqrt((jbp + 1) * (jsp + 1) * (ja + 1) * (kp + 1)) * wigner9j(jb, k, jbp, js, kop, jsp, ja, kp, jap)


@lru_cache(maxsize=None)
def coef_right(jcp, k, jc, jsp, js, kop, jbp, kp, jb):
    """R[c',w,c] += coef * conj(B[c',s',b']) W[w,s',s,w'] R[b',w',b] B[c,s,b]"""
    return (wigner9j(jc, k, jcp, js, kop, jsp, jb, kp, jbp) * sqrt((kp + 1) * (jsp + 1))
            * (jc + 1) * (jbp + 1) / sqrt((jb + 1) * (jcp + 1)))


@lru_cache(maxsize=None)
def coef_apply(ja, jap, k, js1, js1p, kop1, km, jc, jcp, js2, js2p, kop2, kp, jb, jbp):
    """y[a',s1',c',s2',b'] += coef * L[a',w,a] theta[a,s1,c,s2,b] R[b',w',b]^T"""
    return (coef_left(jap, k, ja, js1p, js1, kop1, jcp, km, jc)
            * coef_left(jcp, km, jc, js2p, js2, kop2, jbp, kp, jb) * (jbp + 1) / (jb + 1))


# ----------------------------------------------------------------------------------------------
# sectors, bonds, layouts
# ----------------------------------------------------------------------------------------------
@dataclass
class SiteLayout:
    """layout of a one-site tensor between bond_l and bond_r; kind 'L' groups by the right
    sector (matrix rows = (l, s) groups), kind 'R' groups by the left sector (cols = (s, r))."""
    kind: str
    bond_l: Bond
    bond_r: Bond
    blocks: dict = field(default_factory=dict)     # (l, s, r) -> (off, m, n, ld)
    mats: dict = field(default_factory=dict)       # group sector -> (off, rows, cols, groups)
    size: int = 0

    @staticmethod
    def build(kind, bond_l, bond_r):
        lay = SiteLayout(kind, bond_l, bond_r)
        off = 0
        if kind == "L":
            for r in bond_r:
                groups = [(l, s) for s in range(3) for l in split(r, s) if l in bond_l]
                groups.sort()
                rows = sum(bond_l[l] for (l, s) in groups)
                if rows == 0:
                    continue
                n = bond_r[r]
                ro = 0
                for (l, s) in groups:
                    lay.blocks[(l, s, r)] = (off + ro, bond_l[l], n, rows)
                    ro += bond_l[l]
                lay.mats[r] = (off, rows, n, groups)
                off += rows * n
        else:
            for l in bond_l:
                groups = [(s, r) for s in range(3) for r in fuse(l, s) if r in bond_r]
                groups.sort()
                cols = sum(bond_r[r] for (s, r) in groups)
                if cols == 0:
                    continue
                m = bond_l[l]
                co = 0
                for (s, r) in groups:
                    lay.blocks[(l, s, r)] = (off + co * m, m, bond_r[r], m)
                    co += bond_r[r]
                lay.mats[l] = (off, m, cols, groups)
                off += m * cols
        lay.size = off
        return lay


@dataclass
class ThetaLayout:
    """coupled-sector layout of the two-site tensor between bond_l and bond_r"""
    bond_l: Bond
    bond_r: Bond
    mids: list = field(default_factory=list)
    mats: dict = field(default_factory=dict)      # c -> (off, rows, cols, rowgroups, colgroups, roffs, coffs)
    blocks: dict = field(default_factory=dict)    # (a,s1,c,s2,b) -> (off, m, n, ld)
    size: int = 0

    @staticmethod
    def build(bond_l, bond_r):
        lay = ThetaLayout(bond_l, bond_r)
        rg, cg = {}, {}
        for a in bond_l:
            for s1 in range(3):
                for c in fuse(a, s1):
                    rg.setdefault(c, []).append((a, s1))
        for b in bond_r:
            for s2 in range(3):
                for c in split(b, s2):
                    cg.setdefault(c, []).append((s2, b))
        off = 0
        for c in sorted(set(rg) & set(cg)):
            rows_g = sorted(rg[c])
            cols_g = sorted(cg[c])
            roffs = np.cumsum([0] + [bond_l[a] for (a, _) in rows_g])
            coffs = np.cumsum([0] + [bond_r[b] for (_, b) in cols_g])
            rows, cols = int(roffs[-1]), int(coffs[-1])
            lay.mids.append(c)
            lay.mats[c] = (off, rows, cols, rows_g, cols_g, roffs, coffs)
            for i, (a, s1) in enumerate(rows_g):
                for j, (s2, b) in enumerate(cols_g):
                    lay.blocks[(a, s1, c, s2, b)] = (off + int(roffs[i]) + int(coffs[j]) * rows,
                                                     bond_l[a], bond_r[b], rows)
            off += rows * cols
        lay.size = off
        return lay


@dataclass
class EnvLayout:
    """blocks of a left (side='L': key (bra, w, ket) -> [n_bra, n_ket]) or right
    (side='R': key (ket, w, bra) -> [n_ket, n_bra], i.e. stored transposed) environment."""
    side: str
    bond: Bond
    levels: list                                  # [(dN, k)] of the MPO bond
    blocks: dict = field(default_factory=dict)    # key -> (off, m, n)
    by_ket: dict = field(default_factory=dict)    # (w, ket) -> [bra]
    size: int = 0

    @staticmethod
    def build(side, bond, levels):
        lay = EnvLayout(side, bond, list(levels))
        ident = 0 if side == "L" else len(levels) - 1
        off = 0
        for w, (dN, k) in enumerate(levels):
            if w == ident:
                continue
            for ket in bond:
                for bra in bond:
                    if bra[0] != ket[0] + dN or not triangle(ket[1], k, bra[1]):
                        continue
                    if side == "L":
                        lay.blocks[(bra, w, ket)] = (off, bond[bra], bond[ket])
                    else:
                        lay.blocks[(ket, w, bra)] = (off, bond[ket], bond[bra])
                    lay.by_ket.setdefault((w, ket), []).append(bra)
                    off += bond[bra] * bond[ket]
        lay.size = off
        lay.ident = ident
        return lay


# ----------------------------------------------------------------------------------------------
# task-list assembly
# ----------------------------------------------------------------------------------------------
class TaskList:
    """collects output blocks with their segments and emits the htn_tile / htn_seg arrays"""

    def __init__(self):
        self.blocks = {}     # key -> dict(buf, off, m, n, ld, segs=[...])

    def block(self, key, buf, off, m, n, ld):
        b = self.blocks.get(key)
        if b is None:
            b = dict(buf=buf, off=off, m=m, n=n, ld=ld, segs=[])
            self.blocks[key] = b
        return b

    def gemm(self, key, buf_a, a_off, lda, op_a, buf_b, b_off, ldb, op_b, k, alpha):
        self.blocks[key]["segs"].append((SEG_GEMM, buf_a, a_off, lda, op_a, buf_b, b_off, ldb, op_b, k, alpha))

    def copy(self, key, buf_b, b_off, ldb, alpha):
        self.blocks[key]["segs"].append((SEG_COPY, 0, 0, 1, OP_N, buf_b, b_off, ldb, OP_N, 0, alpha))

    def finalize(self):
        """emit the htn_seg / htn_tile arrays: segment records per block, then ALL tiles in one vectorised pass"""
        seg_rows = []
        blk_rows = []            # (off, buf, ld, m, n, seg_begin, seg_count, ncopy, ksum)
        pos = 0
        flops = 0
        for b in self.blocks.values():
            # merge segments with identical operands (sums alpha)
            merged = {}
            for sg in b["segs"]:
                merged[sg[:10]] = merged.get(sg[:10], 0.0) + sg[10]
            seglist = [(k_, a_) for k_, a_ in merged.items() if a_ != 0.0]
            seglist.sort(key=lambda t: t[0][0])          # GEMM segments first, COPY segments last
            ncopy = 0
            ksum = 0
            for (typ, buf_a, a_off, lda, op_a, buf_b, b_off, ldb, op_b, k), alpha in seglist:
                alpha = complex(alpha)
                seg_rows.append((a_off, b_off, buf_a, buf_b, lda, ldb, k, op_a, op_b, typ, alpha.real, alpha.imag))
                if typ == SEG_GEMM:
                    ksum += k
                else:
                    ncopy += 1
            cnt = len(seglist)
            flops += 8 * b["m"] * b["n"] * ksum
            blk_rows.append((b["off"], b["buf"], b["ld"], b["m"], b["n"], pos, cnt, ncopy, ksum))
            pos += cnt
        segs = np.array(seg_rows, dtype=SEG_DT) if seg_rows else np.zeros(1, dtype=SEG_DT)
        if not blk_rows:
            return Tasks(np.zeros(1, dtype=TILE_DT), 0, segs, pos, flops)
        return _emit_tasks(segs, bool(seg_rows), np.array(blk_rows, dtype=np.int64), pos, flops)


def _emit_tasks(segs, have_segs, B, pos, flops):
    """segment records + block table (off, buf, ld, m, n, seg_begin, seg_count, ncopy, ksum) -> Tasks:
    K pre-split of the GEMM segments and ALL tiles in one vectorised pass"""
    if have_segs:
        # pre-split every GEMM segment into 16-deep K slabs (tile.pad1 = 1): the kernel's quads then walk the
        # slab list without reading segment records, and fetch each slab's descriptor one round ahead
        gem = segs["type"] == SEG_GEMM
        nch = np.where(gem, (segs["k"] + 15) // 16, 1).astype(np.int64)
        cum = np.concatenate([[0], np.cumsum(nch)])
        rep = np.repeat(segs, nch)
        k0 = (np.arange(len(rep)) - np.repeat(cum[:-1], nch)) * 16
        g = rep["type"] == SEG_GEMM
        rep["a_off"] += np.where(g, np.where(rep["op_a"] == OP_N, k0 * rep["lda"], k0), 0)
        rep["b_off"] += np.where(g, np.where(rep["op_b"] == OP_N, k0, k0 * rep["ldb"]), 0)
        rep["k"] = np.where(g, np.minimum(16, rep["k"] - k0), rep["k"])
        B[:, 6] = cum[B[:, 5] + B[:, 6]] - cum[B[:, 5]]
        B[:, 5] = cum[B[:, 5]]
        segs = np.ascontiguousarray(rep)
        pos = len(segs)
    ntr = (B[:, 3] + HTN_TILE - 1) // HTN_TILE
    ntc = (B[:, 4] + HTN_TILE - 1) // HTN_TILE
    nt = ntr * ntc
    bi = np.repeat(np.arange(len(B)), nt)                         # block index of every tile
    first = np.repeat(np.cumsum(nt) - nt, nt)
    loc = np.arange(int(nt.sum())) - first                        # tile index inside its block
    r0 = (loc // ntc[bi]) * HTN_TILE
    c0 = (loc % ntc[bi]) * HTN_TILE
    tarr = np.zeros(len(bi), dtype=TILE_DT)
    tarr["c_off"], tarr["buf_c"], tarr["ldc"] = B[bi, 0], B[bi, 1], B[bi, 2]
    tm = np.minimum(HTN_TILE, B[bi, 3] - r0)
    tn = np.minimum(HTN_TILE, B[bi, 4] - c0)
    tarr["m"], tarr["n"], tarr["row0"], tarr["col0"] = tm, tn, r0, c0
    tarr["seg_begin"], tarr["seg_count"], tarr["pad0"] = B[bi, 5], B[bi, 6], B[bi, 7]
    tarr["pad1"] = 1
    work = tm * tn * (B[bi, 8] + 1)
    tarr = tarr[np.argsort(-work, kind="stable")]                 # longest first: dispatch order = LPT schedule
    return Tasks(np.ascontiguousarray(tarr), len(tarr), segs, pos, flops)


@dataclass
class Tasks:
    tiles: np.ndarray
    ntiles: int
    segs: np.ndarray
    nsegs: int
    flops: int            # algorithmic complex128 flops (8 per MAC) of the GEMM segments


# ----------------------------------------------------------------------------------------------
# plans
# ----------------------------------------------------------------------------------------------
def _w_by_left(W):
    out = {}
    for e in W.entries:
        out.setdefault(e[0], []).append(e)
    return out


def plan_apply(tl: ThetaLayout, Ll: EnvLayout, Rl: EnvLayout, W1, W2):
    """y = H_eff x  (SURVEY 8a a7).  Returns (stageZ Tasks | None, stageY Tasks, z_size, nterms)."""
    nfin = len(W2.right) - 1
    w2 = _w_by_left(W2)
    ty, tz = TaskList(), TaskList()
    for key, (off, m, n, ld) in tl.blocks.items():
        ty.block(key, BUF_Y, off, m, n, ld)
    zoff = 0
    zblocks = {}
    nterms = 0
    for beta, (xoff, xm, xn, xld) in tl.blocks.items():
        a, s1, c, s2, b = beta
        js1, js2 = SITE_MULT[s1][1], SITE_MULT[s2][1]
        for (w, wm, n1, c1) in W1.entries:
            k1, dN1, red1 = SITE_OPS[n1]
            kw, kmid = W1.left[w][1], W1.right[wm][1]
            aps = [a] if w == 0 else Ll.by_ket.get((w, a), [])
            if not aps:
                continue
            for s1p in range(3):
                r1 = red1[s1p, s1]
                if r1 == 0.0:
                    continue
                for (_, wp, n2, c2) in w2.get(wm, []):
                    k2, dN2, red2 = SITE_OPS[n2]
                    kwp = W2.right[wp][1]
                    bps = [b] if wp == nfin else Rl.by_ket.get((wp, b), [])
                    if not bps:
                        continue
                    for s2p in range(3):
                        r2 = red2[s2p, s2]
                        if r2 == 0.0:
                            continue
                        for ap in aps:
                            for cp in fuse(ap, s1p):
                                for bp in bps:
                                    betap = (ap, s1p, cp, s2p, bp)
                                    if betap not in tl.blocks:
                                        continue
                                    cf = coef_apply(a[1], ap[1], kw, js1, SITE_MULT[s1p][1], k1, kmid, c[1],
                                                    cp[1], js2, SITE_MULT[s2p][1], k2, kwp, b[1], bp[1])
                                    alpha = cf * r1 * r2 * c1 * c2
                                    if alpha == 0.0:
                                        continue
                                    nterms += 1
                                    hasL, hasR = (w != 0), (wp != nfin)
                                    if not hasL and not hasR:
                                        ty.copy(betap, BUF_X, xoff, xld, alpha)
                                    elif hasL and not hasR:
                                        lo, lm, ln = Ll.blocks[(ap, w, a)]
                                        ty.gemm(betap, BUF_L, lo, lm, OP_N, BUF_X, xoff, xld, OP_N, ln, alpha)
                                    elif hasR and not hasL:
                                        ro, rm, rn = Rl.blocks[(b, wp, bp)]
                                        ty.gemm(betap, BUF_X, xoff, xld, OP_N, BUF_R, ro, rm, OP_N, rm, alpha)
                                    else:
                                        lo, lm, ln = Ll.blocks[(ap, w, a)]
                                        ro, rm, rn = Rl.blocks[(b, wp, bp)]
                                        zk = (betap, wp, b)
                                        if zk not in zblocks:
                                            zblocks[zk] = zoff
                                            tz.block(zk, BUF_Z, zoff, lm, xn, lm)
                                            ty.gemm(betap, BUF_Z, zoff, lm, OP_N, BUF_R, ro, rm, OP_N, rm, 1.0)
                                            zoff += lm * xn
                                        tz.gemm(zk, BUF_L, lo, lm, OP_N, BUF_X, xoff, xld, OP_N, ln, alpha)
    tasks_z = tz.finalize() if zblocks else None
    return tasks_z, ty.finalize(), zoff, nterms


# ---- dimension-independent ("symbolic") apply plans -------------------------------------------------
# Which blocks meet in which segment, and with which recoupling coefficient, depends only on WHICH sectors
# the two outer bonds hold -- not on their multiplicities.  Near convergence the multiplicities still move
# by a few states from sweep to sweep (every change is a miss of the numeric plan cache, 7-25 ms of Python
# loops here), the sector sets almost never do.  ApplySym keeps the loop-generated structure as index arrays
# into the layouts' block tables; instantiate() turns it into the numeric task lists with a handful of
# numpy gathers (~1 ms).  tests/test_host_cpu.py checks instantiate() == plan_apply() byte for byte.
def _wkey(W):
    k = getattr(W, "_ckey", None)
    if k is None:
        k = (tuple(W.left), tuple(W.right), tuple(W.entries))
        W._ckey = k
    return k


_KIND_C, _KIND_L, _KIND_R, _KIND_Z = 0, 1, 2, 3


class ApplySym:
    __slots__ = ("nterms", "y", "z", "z_first")

    @staticmethod
    def _flatten(per_block, nblocks):
        """per_block: list of {symkey: alpha} in block order -> arrays (blk, kind, i1, i2, alpha) + per-block
        (seg_begin, seg_count, ncopy), GEMM segments first (stable), zero coefficients dropped"""
        blk, kind, i1, i2, al = [], [], [], [], []
        begin = np.zeros(nblocks, dtype=np.int64)
        count = np.zeros(nblocks, dtype=np.int64)
        ncopy = np.zeros(nblocks, dtype=np.int64)
        pos = 0
        for b, d in enumerate(per_block):
            items = [(k, a) for k, a in d.items() if a != 0.0]
            items.sort(key=lambda t: t[0][0] == _KIND_C)         # GEMM segments first, COPY segments last
            begin[b] = pos
            count[b] = len(items)
            for (kd, a1, a2), a in items:
                blk.append(b)
                kind.append(kd)
                i1.append(a1)
                i2.append(a2)
                al.append(a)
                ncopy[b] += kd == _KIND_C
            pos += len(items)
        return (np.array(blk, dtype=np.int64), np.array(kind, dtype=np.int64), np.array(i1, dtype=np.int64),
                np.array(i2, dtype=np.int64), np.array(al, dtype=np.complex128), begin, count, ncopy)

    @staticmethod
    def build(tl, Ll, Rl, W1, W2):
        """the loop nest of plan_apply, recording block INDICES instead of offsets and sizes"""
        nfin = len(W2.right) - 1
        w2 = _w_by_left(W2)
        tidx = {k: i for i, k in enumerate(tl.blocks)}
        lidx = {k: i for i, k in enumerate(Ll.blocks)}
        ridx = {k: i for i, k in enumerate(Rl.blocks)}
        ysegs = [dict() for _ in tidx]
        zsegs, zfirst, zidx = [], [], {}
        nterms = 0
        for beta, xi in tidx.items():
            a, s1, c, s2, b = beta
            js1, js2 = SITE_MULT[s1][1], SITE_MULT[s2][1]
            for (w, wm, n1, c1) in W1.entries:
                k1, dN1, red1 = SITE_OPS[n1]
                kw, kmid = W1.left[w][1], W1.right[wm][1]
                aps = [a] if w == 0 else Ll.by_ket.get((w, a), [])
                if not aps:
                    continue
                for s1p in range(3):
                    r1 = red1[s1p, s1]
                    if r1 == 0.0:
                        continue
                    for (_, wp, n2, c2) in w2.get(wm, []):
                        k2, dN2, red2 = SITE_OPS[n2]
                        kwp = W2.right[wp][1]
                        bps = [b] if wp == nfin else Rl.by_ket.get((wp, b), [])
                        if not bps:
                            continue
                        for s2p in range(3):
                            r2 = red2[s2p, s2]
                            if r2 == 0.0:
                                continue
                            for ap in aps:
                                for cp in fuse(ap, s1p):
                                    for bp in bps:
                                        oi = tidx.get((ap, s1p, cp, s2p, bp))
                                        if oi is None:
                                            continue
                                        cf = coef_apply(a[1], ap[1], kw, js1, SITE_MULT[s1p][1], k1, kmid, c[1],
                                                        cp[1], js2, SITE_MULT[s2p][1], k2, kwp, b[1], bp[1])
                                        alpha = cf * r1 * r2 * c1 * c2
                                        if alpha == 0.0:
                                            continue
                                        nterms += 1
                                        hasL, hasR = (w != 0), (wp != nfin)
                                        d = ysegs[oi]
                                        if not hasL and not hasR:
                                            key = (_KIND_C, xi, 0)
                                            d[key] = d.get(key, 0.0) + alpha
                                        elif hasL and not hasR:
                                            key = (_KIND_L, lidx[(ap, w, a)], xi)
                                            d[key] = d.get(key, 0.0) + alpha
                                        elif hasR and not hasL:
                                            key = (_KIND_R, xi, ridx[(b, wp, bp)])
                                            d[key] = d.get(key, 0.0) + alpha
                                        else:
                                            li, ri = lidx[(ap, w, a)], ridx[(b, wp, bp)]
                                            zk = (oi, wp, b)
                                            zi = zidx.get(zk)
                                            if zi is None:
                                                zi = zidx[zk] = len(zsegs)
                                                zsegs.append({})
                                                zfirst.append((li, xi))
                                                key = (_KIND_Z, zi, ri)
                                                d[key] = d.get(key, 0.0) + 1.0
                                            dz = zsegs[zi]
                                            key = (_KIND_L, li, xi)
                                            dz[key] = dz.get(key, 0.0) + alpha
        sym = ApplySym()
        sym.nterms = nterms
        sym.y = ApplySym._flatten(ysegs, len(tidx))
        sym.z = ApplySym._flatten(zsegs, len(zsegs)) if zsegs else None
        sym.z_first = np.array(zfirst, dtype=np.int64).reshape(-1, 2)
        return sym

    @staticmethod
    def _segs(flat, X, Lb, Rb, zoff, zm):
        blk, kind, i1, i2, al, begin, count, ncopy = flat
        n = len(blk)
        segs = np.zeros(max(n, 1), dtype=SEG_DT)
        if n == 0:
            return segs, np.zeros(len(begin), dtype=np.int64)
        isC, isL, isR, isZ = kind == _KIND_C, kind == _KIND_L, kind == _KIND_R, kind == _KIND_Z
        # operand A
        iL = np.where(isL, i1, 0)
        iXa = np.where(isR, i1, 0)
        iZ = np.where(isZ, i1, 0)
        a_off = np.where(isL, Lb[iL, 0], np.where(isR, X[iXa, 0], np.where(isZ, zoff[iZ], 0)))
        lda = np.where(isL, Lb[iL, 1], np.where(isR, X[iXa, 3], np.where(isZ, zm[iZ], 1)))
        buf_a = np.where(isL, BUF_L, np.where(isR, BUF_X, np.where(isZ, BUF_Z, 0)))
        # operand B
        iXb = np.where(isC, i1, np.where(isL, i2, 0))
        iR = np.where(isR | isZ, i2, 0)
        useX = isC | isL
        b_off = np.where(useX, X[iXb, 0], Rb[iR, 0])
        ldb = np.where(useX, X[iXb, 3], Rb[iR, 1])
        buf_b = np.where(useX, BUF_X, BUF_R)
        k = np.where(isL, Lb[iL, 2], np.where(isR | isZ, Rb[iR, 1], 0))
        segs["a_off"], segs["b_off"], segs["buf_a"], segs["buf_b"] = a_off, b_off, buf_a, buf_b
        segs["lda"], segs["ldb"], segs["k"] = lda, ldb, k
        segs["op_a"], segs["op_b"] = OP_N, OP_N
        segs["type"] = np.where(isC, SEG_COPY, SEG_GEMM)
        segs["alpha_re"], segs["alpha_im"] = al.real, al.imag
        ksum = np.bincount(blk, weights=np.where(isC, 0, k), minlength=len(begin)).astype(np.int64)
        return segs, ksum

    def instantiate(self, tl, Ll, Rl):
        """-> (stageZ Tasks | None, stageY Tasks, z_size, nterms), identical to plan_apply(tl, Ll, Rl, W1, W2)"""
        X = np.array(list(tl.blocks.values()), dtype=np.int64).reshape(-1, 4)
        # (an environment at a chain end holds no blocks: one dummy row keeps the gathers below in range)
        Lb = np.array(list(Ll.blocks.values()) or [(0, 1, 0)], dtype=np.int64).reshape(-1, 3)
        Rb = np.array(list(Rl.blocks.values()) or [(0, 1, 0)], dtype=np.int64).reshape(-1, 3)
        if self.z is not None:
            zm = Lb[self.z_first[:, 0], 1]
            zn = X[self.z_first[:, 1], 2]
            zsz = zm * zn
            zoff = np.cumsum(zsz) - zsz
            zsize = int(zsz.sum())
        else:
            zm = zn = zoff = np.zeros(1, dtype=np.int64)
            zsize = 0
        none = np.zeros(1, dtype=np.int64)
        segs, ksum = ApplySym._segs(self.y, X, Lb, Rb, zoff, zm)
        nb = len(X)
        B = np.stack([X[:, 0], np.full(nb, BUF_Y), X[:, 3], X[:, 1], X[:, 2], self.y[5], self.y[6], self.y[7], ksum], axis=1)
        ny = len(self.y[0])
        ty = _emit_tasks(segs, ny > 0, B.astype(np.int64), ny, int(8 * (X[:, 1] * X[:, 2] * ksum).sum()))
        tz = None
        if self.z is not None:
            segz, ksz = ApplySym._segs(self.z, X, Lb, Rb, none, none)
            Bz = np.stack([zoff, np.full(len(zm), BUF_Z), zm, zm, zn, self.z[5], self.z[6], self.z[7], ksz], axis=1)
            nz = len(self.z[0])
            tz = _emit_tasks(segz, nz > 0, Bz.astype(np.int64), nz, int(8 * (zm * zn * ksz).sum()))
        return tz, ty, zsize, self.nterms


_APPLY_SYM = {}


def plan_apply_cached(tl: ThetaLayout, Ll: EnvLayout, Rl: EnvLayout, W1, W2):
    """plan_apply through the symbolic cache (keyed by the sector SETS of the outer bonds and the MPO sites)"""
    key = (tuple(tl.bond_l.secs), tuple(tl.bond_r.secs), _wkey(W1), _wkey(W2))
    sym = _APPLY_SYM.get(key)
    if sym is None:
        if len(_APPLY_SYM) > 4000:
            _APPLY_SYM.clear()
        sym = _APPLY_SYM[key] = ApplySym.build(tl, Ll, Rl, W1, W2)
    return sym.instantiate(tl, Ll, Rl)


def plan_theta(mode, lay1: SiteLayout, lay2: SiteLayout, tl: ThetaLayout):
    """theta = T1 . T2 on sites (i, i+1).  mode 'RR': both right layout (centre on i);
    'LL': both left layout (centre on i+1); 'LR': left layout x right layout.
    buffers: BUF_S1 = site i, BUF_S2 = site i+1, output BUF_Y."""
    t = TaskList()
    for c in tl.mids:
        off, rows, cols, rows_g, cols_g, roffs, coffs = tl.mats[c]
        if mode == "LR":
            t.block(("m", c), BUF_Y, off, rows, cols, rows)
            if c in lay1.mats and c in lay2.mats:
                o1, r1, n1, g1 = lay1.mats[c]
                o2, m2, c2, g2 = lay2.mats[c]
                assert g1 == rows_g and g2 == cols_g and r1 == rows and c2 == cols
                t.gemm(("m", c), BUF_S1, o1, r1, OP_N, BUF_S2, o2, m2, OP_N, n1, 1.0)
        elif mode == "RR":
            has2 = c in lay2.mats
            if has2:
                o2, m2, c2, g2 = lay2.mats[c]
                assert g2 == cols_g and c2 == cols
            for i, (a, s1) in enumerate(rows_g):
                key = ("r", a, s1, c)
                t.block(key, BUF_Y, off + int(roffs[i]), tl.bond_l[a], cols, rows)
                blk = lay1.blocks.get((a, s1, c))
                if blk is not None and has2:
                    bo, bm, bn, bld = blk
                    t.gemm(key, BUF_S1, bo, bld, OP_N, BUF_S2, o2, m2, OP_N, bn, 1.0)
        elif mode == "LL":
            has1 = c in lay1.mats
            if has1:
                o1, r1, n1, g1 = lay1.mats[c]
                assert g1 == rows_g and r1 == rows
            for j, (s2, b) in enumerate(cols_g):
                key = ("c", c, s2, b)
                t.block(key, BUF_Y, off + int(coffs[j]) * rows, rows, tl.bond_r[b], rows)
                blk = lay2.blocks.get((c, s2, b))
                if blk is not None and has1:
                    bo, bm, bn, bld = blk
                    t.gemm(key, BUF_S1, o1, r1, OP_N, BUF_S2, bo, bld, OP_N, bm, 1.0)
        else:
            raise ValueError(mode)
    return t.finalize()


def plan_left_env(Ll: EnvLayout, lay: SiteLayout, W, Lnew: EnvLayout):
    """GL[i+1] from GL[i], left-layout site tensor (BUF_S1), MPO site W (a10).
    stage 1 -> BUF_Z (Y panels), stage 2 -> BUF_Y (new env).  Returns (t1, t2, z_size)."""
    assert lay.kind == "L"
    t1, t2 = TaskList(), TaskList()
    zoff = 0
    ypanel = {}
    # stage-2 outputs
    for (cp, wp, c), (off, m, n) in Lnew.blocks.items():
        if cp not in lay.mats or c not in lay.mats:
            t2.block((cp, wp, c), BUF_Y, off, m, n, m)      # structurally zero block
            continue
        o_cp, rows_cp, n_cp, g_cp = lay.mats[cp]
        t2.block((cp, wp, c), BUF_Y, off, m, n, m)
        ypanel[(cp, wp, c)] = zoff
        t2.gemm((cp, wp, c), BUF_S1, o_cp, rows_cp, OP_C, BUF_Z, zoff, rows_cp, OP_N, rows_cp, 1.0)
        ro = 0
        for (ap, sp) in g_cp:
            t1.block((cp, wp, c, ap, sp), BUF_Z, zoff + ro, lay.bond_l[ap], n, rows_cp)
            ro += lay.bond_l[ap]
        zoff += rows_cp * n
    for (wl, wr, name, coef) in W.entries:
        if wr == 0 and len(W.right) > 1:
            continue                                  # 'start' level stays the implicit identity
        kop, dN, red = SITE_OPS[name]
        kl, kr = W.left[wl][1], W.right[wr][1]
        for (a, s, c), (aoff, am, an, ald) in lay.blocks.items():
            for sp in range(3):
                r = red[sp, s]
                if r == 0.0:
                    continue
                aps = [a] if wl == 0 else Ll.by_ket.get((wl, a), [])
                for ap in aps:
                    for cp in fuse(ap, sp):
                        if (cp, wr, c) not in ypanel or (ap, sp, cp) not in lay.blocks:
                            continue
                        cf = coef_left(ap[1], kl, a[1], SITE_MULT[sp][1], SITE_MULT[s][1], kop, cp[1], kr, c[1])
                        alpha = cf * r * coef
                        if alpha == 0.0:
                            continue
                        key = (cp, wr, c, ap, sp)
                        if wl == 0:
                            t1.copy(key, BUF_S1, aoff, ald, alpha)
                        else:
                            lo, lm, ln = Ll.blocks[(ap, wl, a)]
                            t1.gemm(key, BUF_L, lo, lm, OP_N, BUF_S1, aoff, ald, OP_N, ln, alpha)
    return t1.finalize(), t2.finalize(), zoff


def plan_right_env(Rl: EnvLayout, lay: SiteLayout, W, Rnew: EnvLayout):
    """GR[i] (stored transposed: block (c, w, c') = [n_c, n_c']) from GR[i+1], right-layout
    site tensor (BUF_S1) and MPO site W.  Returns (t1, t2, z_size)."""
    assert lay.kind == "R"
    t1, t2 = TaskList(), TaskList()
    nfin_r, nfin_l = len(W.right) - 1, len(W.left) - 1
    zoff = 0
    ypanel = {}
    for (c, w, cp), (off, m, n) in Rnew.blocks.items():
        t2.block((c, w, cp), BUF_Y, off, m, n, m)
        if cp not in lay.mats or c not in lay.mats:
            continue
        o_cp, m_cp, cols_cp, g_cp = lay.mats[cp]
        ypanel[(c, w, cp)] = zoff
        # Rt[c, w, c'] (n_c x n_c') = Y (n_c x cols') . B_{c'}^H (cols' x n_c')
        t2.gemm((c, w, cp), BUF_Z, zoff, m, OP_N, BUF_S1, o_cp, m_cp, OP_C, cols_cp, 1.0)
        co = 0
        for (sp, bp) in g_cp:
            t1.block((c, w, cp, sp, bp), BUF_Z, zoff + co * m, m, lay.bond_r[bp], m)
            co += lay.bond_r[bp]
        zoff += m * cols_cp
    for (wl, wr, name, coef) in W.entries:
        if wl == nfin_l and len(W.left) > 1:
            continue                                  # 'final' level stays the implicit identity
        kop, dN, red = SITE_OPS[name]
        kl, kr = W.left[wl][1], W.right[wr][1]
        for (c, s, b), (boff, bm, bn, bld) in lay.blocks.items():
            for sp in range(3):
                r = red[sp, s]
                if r == 0.0:
                    continue
                bps = [b] if wr == nfin_r else Rl.by_ket.get((wr, b), [])
                for bp in bps:
                    for cp in split(bp, sp):
                        if (c, wl, cp) not in ypanel or (cp, sp, bp) not in lay.blocks:
                            continue
                        cf = coef_right(cp[1], kl, c[1], SITE_MULT[sp][1], SITE_MULT[s][1], kop, bp[1], kr, b[1])
                        alpha = cf * r * coef
                        if alpha == 0.0:
                            continue
                        key = (c, wl, cp, sp, bp)
                        if wr == nfin_r:
                            t1.copy(key, BUF_S1, boff, bld, alpha)
                        else:
                            ro, rm, rn = Rl.blocks[(b, wr, bp)]
                            t1.gemm(key, BUF_S1, boff, bld, OP_N, BUF_R, ro, rm, OP_N, rm, alpha)
    return t1.finalize(), t2.finalize(), zoff


# ---- dimension-independent environment-update plans (same idea as ApplySym) -------------------------
class EnvSym:
    """index structure of plan_left_env / plan_right_env; instantiate() fills in offsets and sizes"""
    __slots__ = ("side", "out_has", "out_mat", "zq", "b1_panel", "b1_lb", "s1")

    @staticmethod
    def build(side, Eold, lay, W, Enew):
        left = side == "L"
        assert lay.kind == ("L" if left else "R")
        matidx = {k: i for i, k in enumerate(lay.mats)}
        lbidx = {k: i for i, k in enumerate(lay.blocks)}
        eidx = {k: i for i, k in enumerate(Eold.blocks)}
        sym = EnvSym()
        sym.side = side
        out_has, out_mat, zq = [], [], []
        b1_panel, b1_lb, b1_key = [], [], {}
        ypanel = {}
        for q, key in enumerate(Enew.blocks):
            if left:
                cp, wp, c = key
            else:
                c, wp, cp = key
            has = cp in lay.mats and c in lay.mats
            out_has.append(has)
            out_mat.append(matidx[cp] if has else 0)
            if not has:
                continue
            p_ = len(zq)
            ypanel[key] = p_
            zq.append(q)
            for g in lay.mats[cp][3]:
                if left:
                    (ap, sp) = g
                    k1, lbk = (cp, wp, c, ap, sp), (ap, sp, cp)
                else:
                    (sp, bp) = g
                    k1, lbk = (c, wp, cp, sp, bp), (cp, sp, bp)
                b1_key[k1] = len(b1_panel)
                b1_panel.append(p_)
                b1_lb.append(lbidx[lbk])
        segs = [dict() for _ in b1_panel]
        nfin_r, nfin_l = len(W.right) - 1, len(W.left) - 1
        for (wl, wr, name, coef) in W.entries:
            if left and wr == 0 and len(W.right) > 1:
                continue
            if (not left) and wl == nfin_l and len(W.left) > 1:
                continue
            kop, dN, red = SITE_OPS[name]
            kl, kr = W.left[wl][1], W.right[wr][1]
            for blk, bi in lbidx.items():
                if left:
                    a, s_, c = blk
                else:
                    c, s_, b = blk
                for sp in range(3):
                    r = red[sp, s_]
                    if r == 0.0:
                        continue
                    if left:
                        aps = [a] if wl == 0 else Eold.by_ket.get((wl, a), [])
                        for ap in aps:
                            for cp in fuse(ap, sp):
                                if (cp, wr, c) not in ypanel or (ap, sp, cp) not in lay.blocks:
                                    continue
                                cf = coef_left(ap[1], kl, a[1], SITE_MULT[sp][1], SITE_MULT[s_][1], kop, cp[1], kr, c[1])
                                alpha = cf * r * coef
                                if alpha == 0.0:
                                    continue
                                d = segs[b1_key[(cp, wr, c, ap, sp)]]
                                k = (_KIND_C, bi, 0) if wl == 0 else (_KIND_L, eidx[(ap, wl, a)], bi)
                                d[k] = d.get(k, 0.0) + alpha
                    else:
                        bps = [b] if wr == nfin_r else Eold.by_ket.get((wr, b), [])
                        for bp in bps:
                            for cp in split(bp, sp):
                                if (c, wl, cp) not in ypanel or (cp, sp, bp) not in lay.blocks:
                                    continue
                                cf = coef_right(cp[1], kl, c[1], SITE_MULT[sp][1], SITE_MULT[s_][1], kop, bp[1], kr, b[1])
                                alpha = cf * r * coef
                                if alpha == 0.0:
                                    continue
                                d = segs[b1_key[(c, wl, cp, sp, bp)]]
                                k = (_KIND_C, bi, 0) if wr == nfin_r else (_KIND_R, bi, eidx[(b, wr, bp)])
                                d[k] = d.get(k, 0.0) + alpha
        sym.out_has = np.array(out_has, dtype=bool)
        sym.out_mat = np.array(out_mat, dtype=np.int64)
        sym.zq = np.array(zq, dtype=np.int64)
        sym.b1_panel = np.array(b1_panel, dtype=np.int64)
        sym.b1_lb = np.array(b1_lb, dtype=np.int64)
        sym.s1 = ApplySym._flatten(segs, len(segs))
        return sym

    def instantiate(self, Eold, lay, Enew):
        """-> (t1, t2, z_size), identical to plan_left_env / plan_right_env"""
        left = self.side == "L"
        EN = np.array(list(Enew.blocks.values()) or [(0, 1, 0)], dtype=np.int64).reshape(-1, 3)      # off, m, n
        EO = np.array(list(Eold.blocks.values()) or [(0, 1, 0)], dtype=np.int64).reshape(-1, 3)
        LB = np.array(list(lay.blocks.values()) or [(0, 1, 1, 1)], dtype=np.int64).reshape(-1, 4)    # off, m, n, ld
        MT = np.array([v[:3] for v in lay.mats.values()] or [(0, 1, 1)], dtype=np.int64).reshape(-1, 3)   # off, rows, cols
        nq = len(Enew.blocks)
        zq = self.zq
        npan = len(zq)
        pm = MT[self.out_mat[zq]] if npan else np.zeros((0, 3), dtype=np.int64)
        if left:           # panel: rows_cp x n(new block)
            zrows, zcols = pm[:, 1], EN[zq, 2]
        else:              # panel: m(new block) x cols_cp
            zrows, zcols = EN[zq, 1], pm[:, 2]
        zsz = zrows * zcols
        zoff = np.cumsum(zsz) - zsz
        zsize = int(zsz.sum())
        # ---- stage 2: one GEMM per new block that has a panel ----
        s2 = np.zeros(max(npan, 1), dtype=SEG_DT)
        B2 = np.zeros((nq, 9), dtype=np.int64)
        B2[:, 0], B2[:, 1], B2[:, 2], B2[:, 3], B2[:, 4] = EN[:nq, 0], BUF_Y, EN[:nq, 1], EN[:nq, 1], EN[:nq, 2]
        if npan:
            s2["type"], s2["alpha_re"] = SEG_GEMM, 1.0
            if left:       # new = A_cp^H . Y
                s2["buf_a"], s2["a_off"], s2["lda"], s2["op_a"] = BUF_S1, pm[:, 0], pm[:, 1], OP_C
                s2["buf_b"], s2["b_off"], s2["ldb"], s2["op_b"] = BUF_Z, zoff, pm[:, 1], OP_N
                s2["k"] = pm[:, 1]
            else:          # new = Y . B_cp^H
                s2["buf_a"], s2["a_off"], s2["lda"], s2["op_a"] = BUF_Z, zoff, EN[zq, 1], OP_N
                s2["buf_b"], s2["b_off"], s2["ldb"], s2["op_b"] = BUF_S1, pm[:, 0], pm[:, 1], OP_C
                s2["k"] = pm[:, 2]
            has_pos = np.cumsum(self.out_has) - 1
            B2[:, 5] = np.where(self.out_has, has_pos, npan)        # blocks without a panel: empty segment range
            B2[:, 6] = self.out_has.astype(np.int64)
            B2[:, 8] = np.where(self.out_has, s2["k"][np.clip(has_pos, 0, npan - 1)], 0)
            # TaskList.finalize numbers seg_begin by running position: a block without segments gets the position
            # of the next segment
            B2[:, 5] = np.cumsum(np.concatenate([[0], B2[:-1, 6]]))
        f2 = int(8 * (B2[:, 3] * B2[:, 4] * B2[:, 8]).sum())
        t2 = _emit_tasks(s2, npan > 0, B2, npan, f2) if nq else Tasks(np.zeros(1, dtype=TILE_DT), 0, s2, 0, 0)
        # ---- stage 1: panel sub-blocks ----
        nb1 = len(self.b1_panel)
        pan, lb = self.b1_panel, self.b1_lb
        B1 = np.zeros((nb1, 9), dtype=np.int64)
        if nb1:
            pmat = MT[self.out_mat[zq[pan]]]
            if left:       # rows of group (ap, sp) inside the panel
                B1[:, 0] = zoff[pan] + (LB[lb, 0] - pmat[:, 0])
                B1[:, 2], B1[:, 3], B1[:, 4] = pmat[:, 1], LB[lb, 1], zcols[pan]
            else:          # columns of group (sp, bp): offset co * m
                co = (LB[lb, 0] - pmat[:, 0]) // np.maximum(pmat[:, 1], 1)
                B1[:, 0] = zoff[pan] + co * zrows[pan]
                B1[:, 2], B1[:, 3], B1[:, 4] = zrows[pan], zrows[pan], LB[lb, 2]
            B1[:, 1] = BUF_Z
        blk, kind, i1, i2, al, begin, count, ncopy = self.s1
        n1 = len(blk)
        s1 = np.zeros(max(n1, 1), dtype=SEG_DT)
        ksum = np.zeros(nb1, dtype=np.int64)
        if n1:
            isC, isL, isR = kind == _KIND_C, kind == _KIND_L, kind == _KIND_R
            iE = np.where(isL, i1, np.where(isR, i2, 0))
            iS = np.where(isL, i2, i1)                       # the site block
            s1["type"] = np.where(isC, SEG_COPY, SEG_GEMM)
            s1["alpha_re"], s1["alpha_im"] = al.real, al.imag
            s1["op_a"], s1["op_b"] = OP_N, OP_N
            # COPY: B = site block.  L: A = env (lo, lm), B = site.  R: A = site, B = env (ro, rm), k = rm
            s1["buf_a"] = np.where(isL, BUF_L, np.where(isR, BUF_S1, 0))
            s1["a_off"] = np.where(isL, EO[iE, 0], np.where(isR, LB[iS, 0], 0))
            s1["lda"] = np.where(isL, EO[iE, 1], np.where(isR, LB[iS, 3], 1))
            s1["buf_b"] = np.where(isR, BUF_R, BUF_S1)
            s1["b_off"] = np.where(isR, EO[iE, 0], LB[iS, 0])
            s1["ldb"] = np.where(isR, EO[iE, 1], LB[iS, 3])
            s1["k"] = np.where(isL, EO[iE, 2], np.where(isR, EO[iE, 1], 0))
            ksum = np.bincount(blk, weights=np.where(isC, 0, s1["k"]), minlength=nb1).astype(np.int64)
        B1[:, 5], B1[:, 6], B1[:, 7], B1[:, 8] = begin, count, ncopy, ksum
        f1 = int(8 * (B1[:, 3] * B1[:, 4] * B1[:, 8]).sum())
        t1 = _emit_tasks(s1, n1 > 0, B1, n1, f1) if nb1 else Tasks(np.zeros(1, dtype=TILE_DT), 0, s1, n1, 0)
        return t1, t2, zsize


_ENV_SYM = {}


def plan_env_cached(side, Eold: EnvLayout, lay: SiteLayout, W, Enew: EnvLayout):
    """plan_left_env / plan_right_env through the symbolic cache"""
    key = (side, tuple(lay.bond_l.secs), tuple(lay.bond_r.secs), _wkey(W))
    sym = _ENV_SYM.get(key)
    if sym is None:
        if len(_ENV_SYM) > 4000:
            _ENV_SYM.clear()
        sym = _ENV_SYM[key] = EnvSym.build(side, Eold, lay, W, Enew)
    return sym.instantiate(Eold, lay, Enew)


# ----------------------------------------------------------------------------------------------
# SVD staging / truncation
# ----------------------------------------------------------------------------------------------
@dataclass
class SvdPlan:
    desc: np.ndarray            # SVD_DT per mid sector
    stage: np.ndarray           # COPY_DT items copying M_c or M_c^H into the Jacobi workspace
    mids: list
    transposed: list            # per block: True if G = M^H
    accumulate: list            # per block: True if the rotation J is accumulated (mode B)
    g_size: int
    v_size: int
    s_size: int
    max_m: int
    flops: int                  # LAPACK-equivalent flops, SURVEY 8(d)


def plan_svd(tl: ThetaLayout, placement: str, qrcp: bool = True):
    """Jacobi staging for all coupled blocks.  Default (qrcp): stage G0 = M^H ('right') or M ('left'); the
    kernel preconditions it by pivoted QR and runs Jacobi on R^H, returning (isometry x Sigma) directly
    -- handled downstream exactly like mode A below.  Without qrcp:  The sweep direction decides which isometry is needed
    (placement 'right': U, 'left': V).  Mode A stages the block so that the normalised Jacobi output
    IS that isometry (no rotation accumulated; the centre tensor follows from one GEMM with M);
    mode B (block much wider than tall in that orientation) orthogonalises the short side instead and
    accumulates the rotation J, which then is the wanted isometry."""
    n = len(tl.mids)
    desc = np.zeros(max(n, 1), dtype=SVD_DT)
    stage = np.zeros(max(n, 1), dtype=COPY_DT)
    go = vo = so = 0
    transposed, accumulate = [], []
    max_m = 0
    flops = 0
    for i, c in enumerate(tl.mids):
        off, rows, cols = tl.mats[c][:3]
        mA, nA = (rows, cols) if placement == "right" else (cols, rows)
        if qrcp and max(rows, cols) <= 512:
            m0, n0 = nA, mA                   # G0 is m0 x n0; Jacobi works on R^H: n0 x r
            r = min(m0, n0)
            desc[i] = (go, vo, so, n0, r, 2, m0)
            st = stage[i]
            st["dst_off"], st["src_off"], st["idx_off"], st["scl_off"] = go, off, -1, -1
            st["rows"], st["cols"], st["ldd"], st["lds"] = m0, n0, m0, rows
            st["op"], st["gather_dim"], st["scale_dim"], st["inv_norm"] = (OP_C if placement == "right" else OP_N), 0, -1, 0
            transposed.append(placement != "right")
            accumulate.append(False)
            go += m0 * n0
            vo += ((n0 + 63) // 64 * 64) * r        # padded leading dimension of the kernel's R^H workspace
            so += r
            max_m = max(max_m, m0, n0)
            mm, kk = max(rows, cols), min(rows, cols)
            flops += 4 * (4 * mm * kk * kk + 8 * kk ** 3)
            continue
        modeA = (nA <= 1.25 * mA and mA <= 512) or nA > 512
        if modeA:
            m, nn, tr, acc = mA, nA, placement != "right", False
        else:
            m, nn, tr, acc = nA, mA, placement == "right", True
        if m > 512:
            raise NotImplementedError("coupled block taller than 512 rows in both orientations")
        desc[i] = (go, vo, so, m, nn, 1 if acc else 0, 0)
        st = stage[i]
        st["dst_off"], st["src_off"], st["idx_off"], st["scl_off"] = go, off, -1, -1
        st["rows"], st["cols"], st["ldd"], st["lds"] = m, nn, m, rows
        st["op"], st["gather_dim"], st["scale_dim"], st["inv_norm"] = (OP_C if tr else OP_N), 0, -1, 0
        transposed.append(tr)
        accumulate.append(acc)
        go += m * nn
        vo += nn * nn if acc else 0
        so += nn
        max_m = max(max_m, m)
        mm, kk = max(rows, cols), min(rows, cols)
        flops += 4 * (4 * mm * kk * kk + 8 * kk ** 3)
    return SvdPlan(desc[:max(n, 1)], stage[:max(n, 1)], list(tl.mids), transposed, accumulate, go, vo, so, max_m,
                   flops)


def truncate(svals: dict, chi_full=None, cutoff=0.0, weighting="sqrtdim", rel_floor=1e-14):
    """Global truncation over sectors.  svals: c -> singular values of the
    tilde-normalised block in DESCENDING order.  Schmidt value = s / sqrt(2S+1), each
    (2S+1)-fold degenerate.
      cutoff   -> truncbelow(10^-svalue), src:1007-1010 (keep Schmidt values > cutoff)
      chi_full -> truncdim(D), src:1363-1365 (largest values while sum (2S+1) kept <= D)
    Order at the cut: by sqrt(2S+1) * Schmidt value (= tilde value) for weighting 'sqrtdim', by the Schmidt value
    for 'none'; ties broken by sector label then index, so kept sets are prefixes per sector.
    Returns (keep: c -> count, discarded weight, norm of the kept part)."""
    secs = sorted(svals)
    if not secs:
        return {}, 0.0, 0.0
    vals = np.concatenate([np.asarray(svals[c], dtype=float) for c in secs])
    lens = np.array([len(svals[c]) for c in secs])
    sid = np.repeat(np.arange(len(secs)), lens)
    idx = np.concatenate([np.arange(n) for n in lens]) if len(vals) else np.zeros(0, dtype=int)
    dims = np.array([c[1] + 1 for c in secs], dtype=np.int64)[sid]
    counts, tw, nrm = truncate_arrays(vals, sid, idx, dims, len(secs), chi_full, cutoff, weighting, rel_floor)
    return {c: int(counts[i]) for i, c in enumerate(secs)}, tw, nrm


def truncate_arrays(vals, sid, idx, dims, nsec, chi_full=None, cutoff=0.0, weighting="sqrtdim", rel_floor=1e-14):
    """truncate() on flat arrays: vals sorted descending inside every sector, sid = sector number (sectors numbered
    in sorted label order), idx = position inside the sector, dims = 2S+1 per value.  -> (counts[nsec], discarded
    weight, norm of the kept part)"""
    smax = float(vals.max()) if len(vals) else 0.0
    schmidt = vals / np.sqrt(dims)
    ok = (schmidt > cutoff) & (vals > rel_floor * smax)
    key = vals if weighting == "sqrtdim" else schmidt
    cand = np.nonzero(ok)[0]
    order = cand[np.lexsort((idx[cand], sid[cand], -key[cand]))]       # key desc, then sector, then index
    keep_n = len(order)
    if chi_full is not None and keep_n:
        tot = np.cumsum(dims[order])
        over = np.nonzero(tot > chi_full)[0]
        if len(over):
            keep_n = max(int(over[0]), 1)          # (the largest multiplet is always kept, even if wider than chi_full)
    kept = order[:keep_n]
    counts = np.bincount(sid[kept], minlength=nsec)
    total = float(np.sum(vals ** 2))
    kept_w = float(np.sum(vals[kept] ** 2))
    return counts, (total - kept_w) / total if total > 0 else 0.0, sqrt(kept_w)


def _plan_finalize_fast(tl, sp, order, keep, layA, layB, placement, offA, offB):
    """plan_finalize for the default staging (no accumulated rotations): the same records, filled column-wise
    (this runs once per bond update; the per-block record building of the general path cost 0.5 ms of host time)"""
    right = placement == "right"
    sel = [i for i, c in enumerate(sp.mids) if keep.get(c, 0) > 0]
    nb = len(sel)
    empty = np.zeros(0, dtype=COPY_DT)
    if nb == 0:
        return empty, empty, empty, np.array([0], dtype=np.int32), None
    mids = [sp.mids[i] for i in sel]
    k = np.array([keep[c] for c in mids], dtype=np.int64)
    mats = np.array([tl.mats[c][:3] for c in mids], dtype=np.int64)             # off, rows, cols
    off, rows, cols = mats[:, 0], mats[:, 1], mats[:, 2]
    d = sp.desc[sel]
    g_off, s_off, m = d["g_off"].astype(np.int64), d["s_off"].astype(np.int64), d["m"].astype(np.int64)
    ioff = np.cumsum(k) - k
    idx = np.concatenate([np.asarray(order[c][:kk], dtype=np.int32) for c, kk in zip(mids, k)])
    oA = offA + np.array([layA.mats[c][0] for c in mids], dtype=np.int64)
    oB = offB + np.array([layB.mats[c][0] for c in mids], dtype=np.int64)
    it = np.zeros(nb, dtype=COPY_DT)
    it["idx_off"], it["scl_off"], it["src_off"], it["lds"], it["inv_norm"] = ioff, s_off, g_off, m, 1
    segs = np.zeros(nb, dtype=SEG_DT)
    segs["type"], segs["alpha_re"] = SEG_GEMM, 1.0
    B = np.zeros((nb, 9), dtype=np.int64)       # off, buf, ld, m, n, seg_begin, seg_count, ncopy, ksum
    B[:, 1], B[:, 5], B[:, 6] = BUF_Y, np.arange(nb), 1
    if right:        # G = M, G' = U Sigma: A gathers columns; centre S V^H = U^H M
        it["dst_off"], it["rows"], it["cols"], it["ldd"] = oA, rows, k, rows
        it["gather_dim"], it["op"], it["scale_dim"] = 1, OP_N, 1
        segs["buf_a"], segs["a_off"], segs["lda"], segs["op_a"] = BUF_S1, oA, rows, OP_C
        segs["buf_b"], segs["b_off"], segs["ldb"], segs["op_b"] = BUF_X, off, rows, OP_N
        segs["k"] = rows
        B[:, 0], B[:, 2], B[:, 3], B[:, 4], B[:, 8] = oB, k, k, cols, rows
    else:            # G = M^H, G' = V Sigma: B is the conjugate-transposed gather; centre U S = M V
        it["dst_off"], it["rows"], it["cols"], it["ldd"] = oB, k, cols, k
        it["gather_dim"], it["op"], it["scale_dim"] = 0, OP_C, 0
        segs["buf_a"], segs["a_off"], segs["lda"], segs["op_a"] = BUF_X, off, rows, OP_N
        segs["buf_b"], segs["b_off"], segs["ldb"], segs["op_b"] = BUF_S1, oB, k, OP_C
        segs["k"] = cols
        B[:, 0], B[:, 2], B[:, 3], B[:, 4], B[:, 8] = oA, rows, rows, k, cols
    flops = int(8 * (B[:, 3] * B[:, 4] * B[:, 8]).sum())
    cen_tasks = _emit_tasks(segs, True, B, nb, flops)
    return it, empty, empty, idx.astype(np.int32), cen_tasks


def plan_finalize(tl: ThetaLayout, sp: SvdPlan, order: dict, keep: dict, layA: SiteLayout, layB: SiteLayout,
                  placement: str, offA: int, offB: int):
    """Writes the truncated A (left layout) and B (right layout) into ONE output buffer (A at offA, B at
    offB).  order[c] = permutation sorting the Jacobi column norms descending; keep[c] = kept count.
    Returns (iso_g, cen_g, iso_v copy-item arrays, idx array, centre Tasks | None):
      iso_g : isometry columns taken from G' and divided by sigma           (global scale 1)
      cen_g : centre tensor taken from G' (mode B; sigma cancels)           (global scale 1/nrm)
      iso_v : isometry taken from the accumulated rotation J (mode B)       (global scale 1)
      centre Tasks: mode-A centres, U^H M or M V, as grouped-GEMM segments with alpha = 1 (scaled by the
      caller through alpha_scale) -- buffers: BUF_X = theta, BUF_S1 = BUF_Y = the output buffer."""
    if not any(sp.accumulate):
        return _plan_finalize_fast(tl, sp, order, keep, layA, layB, placement, offA, offB)
    idx = []
    iso_g, cen_g, iso_v = [], [], []
    cen = TaskList()
    right = placement == "right"
    for i, c in enumerate(sp.mids):
        k = keep.get(c, 0)
        if k == 0:
            continue
        off, rows, cols = tl.mats[c][:3]
        g_off, v_off, s_off, m, nn = (int(sp.desc[i][f]) for f in ("g_off", "v_off", "s_off", "m", "n"))
        ioff = len(idx)
        idx.extend(int(p) for p in order[c][:k])
        oA = offA + layA.mats[c][0]
        oB = offB + layB.mats[c][0]
        tr, acc = sp.transposed[i], sp.accumulate[i]
        itA = np.zeros((), dtype=COPY_DT)
        itB = np.zeros((), dtype=COPY_DT)
        # A (rows x k, ld rows): gather columns ; B (k x cols, ld k): conjugate-transposed gather of columns
        itA["dst_off"], itA["rows"], itA["cols"], itA["ldd"] = oA, rows, k, rows
        itA["idx_off"], itA["gather_dim"], itA["op"], itA["scl_off"] = ioff, 1, OP_N, s_off
        itB["dst_off"], itB["rows"], itB["cols"], itB["ldd"] = oB, k, cols, k
        itB["idx_off"], itB["gather_dim"], itB["op"], itB["scl_off"] = ioff, 0, OP_C, s_off
        if right and not acc:            # mode A: G = M, G' = U Sigma
            itA["src_off"], itA["lds"], itA["scale_dim"], itA["inv_norm"] = g_off, m, 1, 1
            iso_g.append(itA)
            key = ("cen", c)
            cen.block(key, BUF_Y, oB, k, cols, k)
            cen.gemm(key, BUF_S1, oA, rows, OP_C, BUF_X, off, rows, OP_N, rows, 1.0)        # U^H M
        elif right and acc:              # mode B: G = M^H, G' = V Sigma, J = U
            itA["src_off"], itA["lds"], itA["scale_dim"], itA["inv_norm"] = v_off, nn, -1, 0
            iso_v.append(itA)
            itB["src_off"], itB["lds"], itB["scale_dim"], itB["inv_norm"] = g_off, m, -1, 0
            cen_g.append(itB)            # S V^H = conj(G')^T
        elif (not right) and not acc:    # mode A: G = M^H, G' = V Sigma
            itB["src_off"], itB["lds"], itB["scale_dim"], itB["inv_norm"] = g_off, m, 0, 1
            iso_g.append(itB)
            key = ("cen", c)
            cen.block(key, BUF_Y, oA, rows, k, rows)
            cen.gemm(key, BUF_X, off, rows, OP_N, BUF_S1, oB, k, OP_C, cols, 1.0)           # M V
        else:                            # mode B: G = M, G' = U Sigma, J = V
            itB["src_off"], itB["lds"], itB["scale_dim"], itB["inv_norm"] = v_off, nn, -1, 0
            iso_v.append(itB)
            itA["src_off"], itA["lds"], itA["scale_dim"], itA["inv_norm"] = g_off, m, -1, 0
            cen_g.append(itA)            # U S = G'

    def arr(lst):
        return np.array(lst, dtype=COPY_DT) if lst else np.zeros(0, dtype=COPY_DT)
    cen_tasks = cen.finalize() if cen.blocks else None
    return arr(iso_g), arr(cen_g), arr(iso_v), np.array(idx if idx else [0], dtype=np.int32), cen_tasks
