"""ORACLE (test infrastructure only -- the product path never imports this).

CPU restatement (numpy, complex128 like the reference, src/HubbardFunctions.jl:264) of the
two-site DMRG hot path that HubbardTN delegates to MPSKit/TensorKit/KrylovKit
(call site src/HubbardFunctions.jl:1010; algorithm summary SURVEY.md App. A.4-A.6):

  * effective two-site Hamiltonian apply  y = GL . theta . O1 . O2 . GR   (a7)
  * Lanczos lowest eigenpair with full reorthogonalisation                 (a8)
  * per-coupled-sector SVD + global truncation (truncdim / truncbelow)     (a9)
  * left / right environment transfer                                      (a10)

on SU(2) x U(1) x fZ2 reduced tensors.  Recoupling coefficients come from brute-force
contraction of explicit Clebsch-Gordan tensors (oracle/su2.py).

The dependencies that hold this algorithm (MPSKit 0.13.1, TensorKit 0.14.6, KrylovKit 0.9.5)
are not vendored in /root/reference and cannot run here; this restatement is pinned by
exact diagonalisation, free-fermion closed forms and exact Schmidt spectra (tests/), and
only loosely (1e-2) by the reference's own test constants.  Parity of truncated runs
against MPSKit itself: UNPINNED.

Data model (all "tilde" normalised, see oracle/su2.py):
  sector          (N, twoS)
  bond            dict sector -> multiplet count
  left tensor  A  dict (a, s, c) -> [n_a, n_c]
  right tensor B  dict (c, s, b) -> [n_c, n_b]
  theta           dict (a, s1, c, s2, b) -> [n_a, n_b]
  envs  L, R      dict (bra sector, level, ket sector) -> matrix, identity levels implicit
"""
from __future__ import annotations

import numpy as np

from . import su2

SITE = su2.SITE_MULT   # ((0,0),(1,1),(2,0))
CDT = np.complex128


# ----------------------------------------------------------------------------------------
# sector bookkeeping
# ----------------------------------------------------------------------------------------
def fuse_site(sec, s):
    """sectors reachable from `sec` by adding site multiplet s"""
    N, j = sec
    Ns, js = SITE[s]
    return [(N + Ns, jj) for jj in su2.couple(j, js)]


def split_site(sec, s):
    """sectors c with  c (x) s -> sec"""
    N, j = sec
    Ns, js = SITE[s]
    if N - Ns < 0:
        return []
    return [(N - Ns, jj) for jj in su2.couple(j, js)]


def full_bonds(L, target):
    """multiplet counts of the exact (untruncated) bond spaces of an L-site chain with total
    sector `target`: paths from the left vacuum intersected with paths to the target."""
    left = [{(0, 0): 1}]
    for i in range(L):
        nxt = {}
        for sec, n in left[-1].items():
            for s in range(3):
                for c in fuse_site(sec, s):
                    nxt[c] = nxt.get(c, 0) + n
        left.append(nxt)
    right = [{target: 1}]
    for i in range(L):
        prv = {}
        for sec, n in right[-1].items():
            for s in range(3):
                for c in split_site(sec, s):
                    prv[c] = prv.get(c, 0) + n
        right.append(prv)
    right = right[::-1]
    bonds = []
    for i in range(L + 1):
        b = {}
        for sec in left[i]:
            if sec in right[i]:
                b[sec] = min(left[i][sec], right[i][sec])
        bonds.append(dict(sorted(b.items())))
    return bonds


def bond_dim_full(bond):
    """TensorKit `dim` of a bond = sum (2S+1) n   (what dim_state prints, src:1399-1405)"""
    return sum((j + 1) * n for (N, j), n in bond.items())


# ----------------------------------------------------------------------------------------
# MPS
# ----------------------------------------------------------------------------------------
class MPS:
    """finite MPS; tensors[i] is the dict of site i+1 (0-based list), kinds[i] in 'L','R'."""

    def __init__(self, L, target):
        self.L, self.target = L, target
        self.bonds = None
        self.tensors = [None] * L
        self.kinds = ["R"] * L


def random_mps(L, target, cap, seed=1234, dtype=CDT):
    """random right-canonical MPS; per-sector multiplet cap `cap` (the analogue of the
    per-sector `bond_dim` cap of initialize_mps, src/HubbardFunctions.jl:931-938)."""
    rng = np.random.default_rng(seed)
    fb = full_bonds(L, target)
    bonds = [{sec: min(n, cap) for sec, n in b.items()} for b in fb]
    psi = MPS(L, target)
    # build right-type tensors from the right end, shrinking bonds where rank requires
    for i in range(L - 1, -1, -1):
        bl, br = bonds[i], bonds[i + 1]
        T = {}
        for c in list(bl):
            cols = [(s, b) for s in range(3) for b in fuse_site(c, s) if b in br]
            ncols = sum(br[b] for (_, b) in cols)
            if ncols == 0:
                del bl[c]
                continue
            nc = min(bl[c], ncols)
            bl[c] = nc
            M = rng.standard_normal((nc, ncols))
            if np.issubdtype(dtype, np.complexfloating):
                M = M + 1j * rng.standard_normal((nc, ncols))
            # orthonormal rows
            q, _ = np.linalg.qr(M.conj().T)
            M = q.conj().T.astype(dtype)
            off = 0
            for (s, b) in cols:
                T[(c, s, b)] = M[:, off:off + br[b]].copy()
                off += br[b]
        psi.tensors[i] = T
    # prune sectors of bond i+1 that lost all their support on the left
    for i in range(L):
        bl, br = bonds[i], bonds[i + 1]
        T = psi.tensors[i]
        for key in [k for k in T if k[0] not in bl]:
            del T[key]
    psi.bonds = bonds
    assert bonds[0] == {(0, 0): 1}, bonds[0]
    return psi


# ----------------------------------------------------------------------------------------
# environments
# ----------------------------------------------------------------------------------------
def _ops():
    return su2.site_operators()


def left_env_step(Lenv, A, W, bond_l, bond_r):
    """GL[i+1] = sum GL[i] . A . O . conj(A)   (a10); A left-type dict of this site.
    Lenv: dict (a', w, a) -> matrix [n_a', n_a]; level `start` is implicit identity (absent)."""
    ops = _ops()
    start_l = 0
    out = {}
    for (wl, wr, name, coef) in W["entries"]:
        kop, dN, red = ops[name]
        dNl, kl = W["left"][wl]
        dNr, kr = W["right"][wr]
        is_start_out = (wr == 0 and len(W["right"]) > 1 and name == "id" and wl == 0)
        if is_start_out:
            continue  # start -> start stays implicit identity
        for (b, s, a), Ablk in A.items():
            for sp in range(3):
                r = red[sp, s]
                if r == 0.0:
                    continue
                # bra side: b' from L block
                if wl == start_l and _is_identity_level(W, "left", wl):
                    bps = [b]
                else:
                    bps = [bp for (bp, w, bb) in Lenv if w == wl and bb == b]
                for bp in bps:
                    for ap in fuse_site(bp, sp):
                        if (bp, sp, ap) not in A:
                            continue
                        if ap[0] != a[0] + dNr or not su2.triangle(a[1], kr, ap[1]):
                            continue
                        cf = su2.coef_left_env(bp[1], kl, b[1], SITE[sp][1], SITE[s][1], kop,
                                               ap[1], kr, a[1])
                        if cf == 0.0:
                            continue
                        Lb = None if (wl == start_l and _is_identity_level(W, "left", wl)) else Lenv[(bp, wl, b)]
                        X = Ablk if Lb is None else Lb @ Ablk            # [n_b', n_a]
                        Y = A[(bp, sp, ap)].conj().T @ X                   # [n_a', n_a]
                        key = (ap, wr, a)
                        if key in out:
                            out[key] = out[key] + (cf * r * coef) * Y
                        else:
                            out[key] = (cf * r * coef) * Y
    return out


def _is_identity_level(W, side, w):
    """level 0 of an interior/left-boundary bond is the untouched identity ('start');
    the last level of a right bond is 'final' whose right environment is the identity."""
    if side == "left":
        return w == 0
    return w == len(W["right"]) - 1


def right_env_step(Renv, B, W, bond_l, bond_r):
    """GR[i] = sum B . O . GR[i+1] . conj(B); B right-type dict.  `final` level implicit identity."""
    ops = _ops()
    out = {}
    nfin_r = len(W["right"]) - 1
    nfin_l = len(W["left"]) - 1
    for (wl, wr, name, coef) in W["entries"]:
        kop, dN, red = ops[name]
        dNl, kl = W["left"][wl]
        dNr, kr = W["right"][wr]
        if wl == nfin_l and wr == nfin_r and name == "id" and len(W["left"]) > 1:
            continue  # final -> final stays implicit identity
        r_ident = (wr == nfin_r)
        for (c, s, b), Bblk in B.items():
            for sp in range(3):
                r = red[sp, s]
                if r == 0.0:
                    continue
                if r_ident:
                    bps = [b]
                else:
                    bps = [bp for (bp, w, bb) in Renv if w == wr and bb == b]
                for bp in bps:
                    for cp in split_site(bp, sp):
                        if (cp, sp, bp) not in B:
                            continue
                        if cp[0] != c[0] + dNl or not su2.triangle(c[1], kl, cp[1]):
                            continue
                        cf = su2.coef_right_env(cp[1], kl, c[1], SITE[sp][1], SITE[s][1], kop,
                                                bp[1], kr, b[1])
                        if cf == 0.0:
                            continue
                        X = Bblk if r_ident else Bblk @ Renv[(bp, wr, b)].T   # [n_c, n_b']
                        Y = B[(cp, sp, bp)].conj() @ X.T                        # [n_c', n_c]
                        key = (cp, wl, c)
                        if key in out:
                            out[key] = out[key] + (cf * r * coef) * Y
                        else:
                            out[key] = (cf * r * coef) * Y
    return out


# ----------------------------------------------------------------------------------------
# two-site effective Hamiltonian
# ----------------------------------------------------------------------------------------
def theta_blocks(bond_l, bond_r):
    """all (a, s1, c, s2, b) allowed by fusion with a in bond_l and b in bond_r"""
    out = []
    for a in bond_l:
        for s1 in range(3):
            for c in fuse_site(a, s1):
                for s2 in range(3):
                    for b in fuse_site(c, s2):
                        if b in bond_r:
                            out.append((a, s1, c, s2, b))
    return out


def build_apply_terms(blocks, Lenv, Renv, W1, W2):
    """list of (beta_out, beta_in, Lkey|None, Rkey|None, coef): the task list of one H_eff apply"""
    ops = _ops()
    blockset = set(blocks)
    nfin = len(W2["right"]) - 1
    by_left = {}
    for e2 in W2["entries"]:
        by_left.setdefault(e2[0], []).append(e2)
    Lidx = {}
    for (ap, w, a) in Lenv:
        Lidx.setdefault((w, a), []).append(ap)
    Ridx = {}
    for (bp, w, b) in Renv:
        Ridx.setdefault((w, b), []).append(bp)
    terms = {}
    for beta in blocks:
        a, s1, c, s2, b = beta
        for (w, wm, n1, c1) in W1["entries"]:
            k1, dN1, red1 = ops[n1]
            kw = W1["left"][w][1]
            kmid = W1["right"][wm][1]
            aps = [a] if w == 0 else Lidx.get((w, a), [])
            if not aps:
                continue
            for s1p in range(3):
                r1 = red1[s1p, s1]
                if r1 == 0.0:
                    continue
                for (_, wp, n2, c2) in by_left.get(wm, []):
                    k2, dN2, red2 = ops[n2]
                    kwp = W2["right"][wp][1]
                    bps = [b] if wp == nfin else Ridx.get((wp, b), [])
                    if not bps:
                        continue
                    for s2p in range(3):
                        r2 = red2[s2p, s2]
                        if r2 == 0.0:
                            continue
                        for ap in aps:
                            for cp in fuse_site(ap, s1p):
                                for bp in bps:
                                    betap = (ap, s1p, cp, s2p, bp)
                                    if betap not in blockset:
                                        continue
                                    cf = su2.coef_apply(a[1], ap[1], kw, SITE[s1][1], SITE[s1p][1], k1,
                                                        kmid, c[1], cp[1], SITE[s2][1], SITE[s2p][1],
                                                        k2, kwp, b[1], bp[1])
                                    if cf == 0.0:
                                        continue
                                    Lk = None if w == 0 else (ap, w, a)
                                    Rk = None if wp == nfin else (bp, wp, b)
                                    key = (betap, beta, Lk, Rk)
                                    terms[key] = terms.get(key, 0.0) + cf * r1 * r2 * c1 * c2
    return [(k[0], k[1], k[2], k[3], v) for k, v in terms.items() if v != 0.0]


def apply_heff(theta, terms, Lenv, Renv):
    y = {k: np.zeros_like(v) for k, v in theta.items()}
    for (bo, bi, Lk, Rk, cf) in terms:
        X = theta[bi]
        if Lk is not None:
            X = Lenv[Lk] @ X
        if Rk is not None:
            X = X @ Renv[Rk].T
        y[bo] += cf * X
    return y


# ----------------------------------------------------------------------------------------
# Lanczos (KrylovKit-style: keep all vectors, full reorthogonalisation, eager stop; A.5)
# ----------------------------------------------------------------------------------------
def _flat(theta, order):
    return np.concatenate([theta[k].ravel() for k in order])


def _unflat(vec, order, shapes):
    out, off = {}, 0
    for k in order:
        n = shapes[k][0] * shapes[k][1]
        out[k] = vec[off:off + n].reshape(shapes[k])
        off += n
    return out


def lanczos_lowest(matvec, x0, krylovdim=30, tol=1e-12, maxrestart=10):
    """lowest eigenpair of the Hermitian map `matvec` on flat vectors.  Returns
    (eigval, eigvec, n_matvec, residual)."""
    x = x0 / np.linalg.norm(x0)
    nmv = 0
    for _ in range(maxrestart + 1):
        V = [x]
        alphas, betas = [], []
        theta_val, y, res = None, None, None
        for j in range(krylovdim):
            w = matvec(V[j])
            nmv += 1
            a = np.vdot(V[j], w).real
            alphas.append(a)
            w = w - a * V[j]
            if j > 0:
                w = w - betas[j - 1] * V[j - 1]
            # full reorthogonalisation, two passes
            for _p in range(2):
                Vm = np.array(V)
                w = w - Vm.T @ (Vm.conj() @ w)
            bnorm = np.linalg.norm(w)
            T = np.diag(alphas) + np.diag(betas, 1) + np.diag(betas, -1)
            ev, evec = np.linalg.eigh(T)
            theta_val, y = ev[0], evec[:, 0]
            res = abs(bnorm * y[-1])
            if res < tol or bnorm < 1e-14 or j == krylovdim - 1:
                break
            betas.append(bnorm)
            V.append(w / bnorm)
        Vm = np.array(V[:len(y)])
        x = Vm.T @ y
        x = x / np.linalg.norm(x)
        if res < tol or bnorm < 1e-14:
            break
    return theta_val, x, nmv, res


# ----------------------------------------------------------------------------------------
# SVD + truncation (a9, App. A.6)
# ----------------------------------------------------------------------------------------
def coupled_blocks(theta, bond_l, bond_r):
    """group theta by the mid sector c: c -> (rows [(a,s1)], cols [(s2,b)], dense matrix)"""
    groups = {}
    for (a, s1, c, s2, b) in theta:
        g = groups.setdefault(c, (set(), set()))
        g[0].add((a, s1))
        g[1].add((s2, b))
    out = {}
    for c, (rs, cs) in groups.items():
        rows = sorted(rs)
        cols = sorted(cs)
        roff = np.cumsum([0] + [bond_l[a] for (a, _) in rows])
        coff = np.cumsum([0] + [bond_r[b] for (_, b) in cols])
        dt = next(iter(theta.values())).dtype
        M = np.zeros((roff[-1], coff[-1]), dtype=dt)
        for i, (a, s1) in enumerate(rows):
            for k, (s2, b) in enumerate(cols):
                blk = theta.get((a, s1, c, s2, b))
                if blk is not None:
                    M[roff[i]:roff[i + 1], coff[k]:coff[k + 1]] = blk
        out[c] = (rows, cols, roff, coff, M)
    return out


def truncate_spectrum(svals, chi_full=None, cutoff=0.0, weighting="sqrtdim"):
    """global truncation over sectors.  svals: c -> descending singular values of the
    tilde-normalised block (Schmidt value = s / sqrt(2S+1), degeneracy 2S+1).

    truncbelow(eta) (src:1010): keep Schmidt values > eta.
    truncdim(D)     (src:1363-1365): keep the largest values while sum (2S+1) kept <= D.
      TensorKit's ordering weight at the cut is unverifiable here (App. A.6 flag):
      weighting='sqrtdim' orders by (2S+1)^(1/2) * schmidt = s (tilde value),
      weighting='none'    orders by the Schmidt value s / sqrt(2S+1).
    Returns c -> number kept, and the discarded weight sum s^2 (state normalised to 1).
    """
    items = []
    smax = max((float(s[0]) for s in svals.values() if len(s)), default=0.0)
    for c, s in svals.items():
        d = c[1] + 1
        for i, v in enumerate(s):
            schmidt = v / np.sqrt(d)
            if schmidt <= cutoff or v <= 1e-14 * smax:     # numerically zero directions are never kept
                continue
            key = v if weighting == "sqrtdim" else schmidt
            items.append((key, c, i, d))
    # stable order: by key desc, then sector, then index
    items.sort(key=lambda t: (-t[0], t[1], t[2]))
    keep = {c: 0 for c in svals}
    tot = 0
    for key, c, i, d in items:
        if chi_full is not None and tot + d > chi_full and tot > 0:
            break           # stop at the first multiplet that does not fit (contiguous prefix)
        # values within a sector arrive in descending order, so kept sets are prefixes
        keep[c] += 1
        tot += d
    total = sum(float(np.sum(s ** 2)) for s in svals.values())
    kept = sum(float(np.sum(svals[c][:keep[c]] ** 2)) for c in svals)
    return keep, (total - kept) / total


def svd_truncate(theta, bond_l, bond_r, chi_full=None, cutoff=0.0, weighting="sqrtdim"):
    """theta -> (A left-type, S per sector, B right-type, new mid bond, trunc weight, spectrum)"""
    cb = coupled_blocks(theta, bond_l, bond_r)
    fac = {}
    svals = {}
    for c, (rows, cols, roff, coff, M) in cb.items():
        U, s, Vh = np.linalg.svd(M, full_matrices=False)
        fac[c] = (U, s, Vh)
        svals[c] = s
    keep, tw = truncate_spectrum(svals, chi_full, cutoff, weighting)
    nrm = np.sqrt(sum(float(np.sum(svals[c][:keep[c]] ** 2)) for c in svals))
    A, B, S, mid = {}, {}, {}, {}
    for c, (rows, cols, roff, coff, M) in cb.items():
        k = keep[c]
        if k == 0:
            continue
        U, s, Vh = fac[c]
        mid[c] = k
        S[c] = s[:k] / nrm
        for i, (a, s1) in enumerate(rows):
            A[(a, s1, c)] = U[roff[i]:roff[i + 1], :k].copy()
        for j, (s2, b) in enumerate(cols):
            B[(c, s2, b)] = Vh[:k, coff[j]:coff[j + 1]].copy()
    mid = dict(sorted(mid.items()))
    return A, S, B, mid, tw, svals


# ----------------------------------------------------------------------------------------
# sweep driver (finite DMRG2, App. A.4)
# ----------------------------------------------------------------------------------------
class DMRG2:
    def __init__(self, psi: MPS, mpo, chi_full=None, cutoff=0.0, krylovdim=30, lanczos_tol=1e-12,
                 weighting="sqrtdim", maxrestart=3):
        self.psi, self.mpo = psi, mpo
        self.L = psi.L
        self.chi_full, self.cutoff = chi_full, cutoff
        self.krylovdim, self.lanczos_tol, self.maxrestart = krylovdim, lanczos_tol, maxrestart
        self.weighting = weighting
        self.Lenvs = [None] * (self.L + 1)
        self.Renvs = [None] * (self.L + 1)
        self.Lenvs[0] = {}
        self.Renvs[self.L] = {}
        # right environments for a right-canonical start
        for i in range(self.L - 1, 0, -1):
            self.Renvs[i] = right_env_step(self.Renvs[i + 1], psi.tensors[i], mpo[i],
                                           psi.bonds[i], psi.bonds[i + 1])
        self.center = None       # dict c -> S on bond 1 when starting
        self.energy = None
        self.stats = []

    # -- one bond update; sites (i, i+1) 0-based i ------------------------------------
    def _theta(self, i, direction):
        psi = self.psi
        T1, T2 = psi.tensors[i], psi.tensors[i + 1]
        bl, br = psi.bonds[i], psi.bonds[i + 2]
        theta = {}
        for (a, s1, c), X in T1.items():
            for s2 in range(3):
                for b in fuse_site(c, s2):
                    Y = T2.get((c, s2, b))
                    if Y is None:
                        continue
                    theta[(a, s1, c, s2, b)] = X @ Y
        # complete with structurally allowed zero blocks so new sectors can appear
        for beta in theta_blocks(bl, br):
            if beta not in theta:
                a, _, _, _, b = beta
                theta[beta] = np.zeros((bl[a], br[b]), dtype=CDT)
        return theta

    def update_bond(self, i, direction, placement=None):
        """placement 'right': A_i = U, centre S V^H on site i+1; 'left': centre U S on i, B_{i+1} = V^H"""
        if placement is None:
            placement = "right" if direction > 0 else "left"
        psi = self.psi
        bl, br = psi.bonds[i], psi.bonds[i + 2]
        theta = self._theta(i, direction)
        blocks = sorted(theta)
        Lenv, Renv = self.Lenvs[i], self.Renvs[i + 2]
        terms = build_apply_terms(blocks, Lenv, Renv, self.mpo[i], self.mpo[i + 1])
        shapes = {k: theta[k].shape for k in blocks}

        def mv(vec):
            y = apply_heff(_unflat(vec, blocks, shapes), terms, Lenv, Renv)
            return _flat(y, blocks)

        x0 = _flat(theta, blocks)
        E, x, nmv, res = lanczos_lowest(mv, x0, self.krylovdim, self.lanczos_tol, self.maxrestart)
        theta = _unflat(x, blocks, shapes)
        A, S, B, mid, tw, svals = svd_truncate(theta, bl, br, self.chi_full, self.cutoff, self.weighting)
        psi.bonds[i + 1] = mid
        if placement == "right":
            psi.tensors[i] = A
            psi.kinds[i] = "L"
            psi.tensors[i + 1] = {k: S[k[0]][:, None] * v for k, v in B.items()}   # centre on site i+1
            self.Lenvs[i + 1] = left_env_step(Lenv, A, self.mpo[i], bl, mid)
        else:
            psi.tensors[i + 1] = B
            psi.kinds[i + 1] = "R"
            psi.tensors[i] = {k: v * S[k[2]][None, :] for k, v in A.items()}       # centre on site i
            self.Renvs[i + 1] = right_env_step(Renv, B, self.mpo[i + 1], mid, br)
        self.energy = E
        self.stats.append(dict(bond=i + 1, dir=direction, E=E, nmv=nmv, res=res, trunc=tw,
                               chi_full=bond_dim_full(mid), mult=sum(mid.values()),
                               nterms=len(terms), nblocks=len(blocks),
                               size=sum(v.size for v in theta.values())))
        return E, {c: S[c] / np.sqrt(c[1] + 1) for c in S}

    def sweep(self):
        """one sweep in MPSKit's DMRG2 order (App. A.4): bonds 1..L-1 rightwards, then L-2..1
        leftwards = 2L-3 bond updates; the last rightward update leaves the centre on the left."""
        spectra = {}
        L = self.L
        for i in range(L - 1):
            E, sp = self.update_bond(i, +1, "right" if i < L - 2 else "left")
            spectra[i + 1] = sp
        for i in range(L - 3, -1, -1):
            E, sp = self.update_bond(i, -1, "left")
            spectra[i + 1] = sp
        return self.energy, spectra
