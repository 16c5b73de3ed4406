"""ORACLE (test infrastructure only -- never imported by the product path).

SU(2) building blocks "by definition" (SURVEY.md App. A.8): explicit Clebsch-Gordan
tensors, structure tensors of every object on the two-site DMRG path, and recoupling
coefficients obtained by brute-force contraction of those structure tensors.  No 6j/9j
closed form is trusted here; the product planner's closed forms are tested against
these numbers.

All spins are stored doubled (twoS ints).  Component index idx = 0..twoS maps to
twoM = twoS - 2*idx (highest weight first), so for spin 1/2: idx 0 = up, idx 1 = down.

Reference semantics restated here:
  * fZ2 x SU(2) x U(1) sectors, src/HubbardFunctions.jl:250-251 (we carry the particle
    number N instead of the shifted charge k = N*Q - P*sites; parity = N mod 2).
  * reduced matrix elements of c^+, c (1, sqrt 2), src/HubbardFunctions.jl:281-293.
  * quantum-dimension weighted inner product of TensorKit -- absorbed
    into the "tilde" normalisation below so that all reduced data live in a plain
    Euclidean metric.
"""
from __future__ import annotations

from functools import lru_cache
from math import factorial, sqrt

import numpy as np


# ----------------------------------------------------------------------------------------
# Clebsch-Gordan coefficients (Racah's formula, doubled arguments)
# ----------------------------------------------------------------------------------------
def _f(n2: int) -> int:
    """factorial of n2/2 for an even, non-negative doubled argument"""
    assert n2 >= 0 and n2 % 2 == 0, n2
    return factorial(n2 // 2)


@lru_cache(maxsize=None)
def cg(j1: int, m1: int, j2: int, m2: int, j: int, m: int) -> float:
    """<j1 m1; j2 m2 | j m>, all arguments doubled (Condon-Shortley phases)."""
    if m1 + m2 != m:
        return 0.0
    if j < abs(j1 - j2) or j > j1 + j2 or (j1 + j2 + j) % 2:
        return 0.0
    if abs(m1) > j1 or abs(m2) > j2 or abs(m) > j:
        return 0.0
    if (j1 + m1) % 2 or (j2 + m2) % 2 or (j + m) % 2:
        return 0.0
    pref = (j + 1) * _f(j + j1 - j2) * _f(j - j1 + j2) * _f(j1 + j2 - j) / _f(j1 + j2 + j + 2)
    pref *= _f(j + m) * _f(j - m) * _f(j1 - m1) * _f(j1 + m1) * _f(j2 - m2) * _f(j2 + m2)
    s = 0.0
    # k runs over integers (not doubled)
    kmin = max(0, (j2 - j - m1) // 2, (j1 + m2 - j) // 2)
    kmax = min((j1 + j2 - j) // 2, (j1 - m1) // 2, (j2 + m2) // 2)
    for k in range(kmin, kmax + 1):
        den = (factorial(k) * factorial((j1 + j2 - j) // 2 - k) * factorial((j1 - m1) // 2 - k)
               * factorial((j2 + m2) // 2 - k) * factorial((j - j2 + m1) // 2 + k)
               * factorial((j - j1 - m2) // 2 + k))
        s += (-1) ** k / den
    return sqrt(pref) * s


@lru_cache(maxsize=None)
def cg_tensor(j1: int, j2: int, j: int) -> np.ndarray:
    """C[i1, i2, i] = <j1 m1; j2 m2 | j m> with idx -> twoM = twoJ - 2 idx."""
    out = np.zeros((j1 + 1, j2 + 1, j + 1))
    for i1 in range(j1 + 1):
        for i2 in range(j2 + 1):
            for i in range(j + 1):
                out[i1, i2, i] = cg(j1, j1 - 2 * i1, j2, j2 - 2 * i2, j, j - 2 * i)
    out.setflags(write=False)
    return out


def triangle(j1: int, j2: int, j: int) -> bool:
    return abs(j1 - j2) <= j <= j1 + j2 and (j1 + j2 + j) % 2 == 0


def couple(j1: int, j2: int):
    """all j in j1 (x) j2 (doubled)"""
    return range(abs(j1 - j2), j1 + j2 + 1, 2)


# ----------------------------------------------------------------------------------------
# Local Hilbert space of one spinful fermion site
# ----------------------------------------------------------------------------------------
# full basis: |0>, |up>, |dn>, |updn> := c+_up c+_dn |0>   (mode order: up before down)
# multiplets sigma = 0,1,2 : (N, twoS) = (0,0), (1,1), (2,0)
SITE_MULT = ((0, 0), (1, 1), (2, 0))
SITE_OFFSET = (0, 1, 3)          # first full-basis index of each multiplet
D_FULL = 4


def local_matrices():
    """Jordan-Wigner local 4x4 matrices: a_up, a_dn (annihilators incl. the intra-site
    string), F = (-1)^n, n, n_up n_dn.  Only used to *define* reduced operators and to
    build dense Hamiltonians for checks."""
    a_up = np.zeros((4, 4))
    a_dn = np.zeros((4, 4))
    # a_up |up> = |0>, a_up |updn> = |dn>
    a_up[0, 1] = 1.0
    a_up[2, 3] = 1.0
    # a_dn |dn> = |0>, a_dn |updn> = a_dn c+_up c+_dn|0> = - c+_up |0> = -|up>
    a_dn[0, 2] = 1.0
    a_dn[1, 3] = -1.0
    n = np.diag([0.0, 1.0, 1.0, 2.0])
    F = np.diag([1.0, -1.0, -1.0, 1.0])
    docc = np.diag([0.0, 0.0, 0.0, 1.0])
    return dict(a_up=a_up, a_dn=a_dn, F=F, n=n, docc=docc, id=np.eye(4))


def reduce_site_operator(comps, k: int):
    """Wigner-Eckart reduction of a site tensor operator.

    comps[q_idx] is the 4x4 matrix of component twoQ = k - 2 q_idx.  Returns
    red[sigma_out, sigma_in] with  <s' m'| T_q |s m> = red[s', s] * <j_s m; k q | j_s' m'>
    and asserts that the set really is an irreducible tensor operator of rank k.
    """
    red = np.zeros((3, 3))
    for so, (No, jo) in enumerate(SITE_MULT):
        for si, (Ni, ji) in enumerate(SITE_MULT):
            if not triangle(ji, k, jo):
                # all matrix elements must vanish
                for q in range(k + 1):
                    blk = comps[q][SITE_OFFSET[so]:SITE_OFFSET[so] + jo + 1,
                                   SITE_OFFSET[si]:SITE_OFFSET[si] + ji + 1]
                    assert np.allclose(blk, 0), "not a tensor operator"
                continue
            C = cg_tensor(ji, k, jo)          # [mi, q, mo]
            full = np.zeros((ji + 1, k + 1, jo + 1))
            for q in range(k + 1):
                blk = comps[q][SITE_OFFSET[so]:SITE_OFFSET[so] + jo + 1,
                               SITE_OFFSET[si]:SITE_OFFSET[si] + ji + 1]   # [mo, mi]
                full[:, q, :] = blk.T
            r = float(np.sum(full * C) / np.sum(C * C))
            assert np.allclose(full, r * C, atol=1e-13), "not a rank-%d tensor operator" % k
            red[so, si] = r
    return red


@lru_cache(maxsize=None)
def site_operators():
    """Named reduced site operators: name -> (k, dN, red[3,3]).

    Emit/absorb split follows the Jordan-Wigner form of a hopping term
        c+_{i s} c_{j s} = (a+_s F)_i  F_{i+1..j-1}  (a_s)_j          (i < j)
        c+_{j s} c_{i s} = (F a_s)_i   F_{i+1..j-1}  (a+_s)_j
    The annihilator components are arranged as the conjugate spinor
        ctil_{+1/2} = -a_dn , ctil_{-1/2} = a_up
    so that sum_q <1/2 q; 1/2 -q|0 0> c+_q ctil_{-q} = (1/sqrt 2) sum_s c+_s c_s.
    Reduced values reproduce src/HubbardFunctions.jl:284-290 up to the fusion-tensor
    normalisation (1 and sqrt 2 there).
    """
    lm = local_matrices()
    a_up, a_dn, F = lm["a_up"], lm["a_dn"], lm["F"]
    ops = {}
    ops["id"] = (0, 0, reduce_site_operator([lm["id"]], 0))
    ops["F"] = (0, 0, reduce_site_operator([F], 0))
    ops["n"] = (0, 0, reduce_site_operator([lm["n"]], 0))
    ops["docc"] = (0, 0, reduce_site_operator([lm["docc"]], 0))
    ops["nF"] = (0, 0, reduce_site_operator([lm["n"] @ F], 0))                  # the density inside a Jordan-Wigner string
    # creators: components (up, down)
    ops["cdag"] = (1, +1, reduce_site_operator([a_up.T, a_dn.T], 1))          # absorb side
    ops["cdagF"] = (1, +1, reduce_site_operator([a_up.T @ F, a_dn.T @ F], 1))  # emit side
    # annihilators as conjugate spinor (-a_dn, a_up)
    ops["c"] = (1, -1, reduce_site_operator([-a_dn, a_up], 1))                 # absorb side
    ops["Fc"] = (1, -1, reduce_site_operator([-F @ a_dn, F @ a_up], 1))        # emit side
    # spin operator as a rank-1 spherical tensor (components q = +1, 0, -1) and on-site pair operators; all even
    # under fermion parity, so they need no Jordan-Wigner string.  Used by the exchange terms (src:426-428, 445-451)
    sp = a_up.T @ a_dn                          # S^+
    sz = 0.5 * (a_up.T @ a_up - a_dn.T @ a_dn)
    ops["S"] = (2, 0, reduce_site_operator([-sp / np.sqrt(2.0), sz, sp.T / np.sqrt(2.0)], 2))
    ops["pair_dag"] = (0, +2, reduce_site_operator([a_up.T @ a_dn.T], 0))       # c+_up c+_dn
    ops["pair"] = (0, -2, reduce_site_operator([a_dn @ a_up], 0))               # c_dn c_up
    # density-assisted ladder operators n_{-s} c+_s / n_{-s} c_s of the three-equal-index terms (src:429-433, 452-458)
    n_up, n_dn = a_up.T @ a_up, a_dn.T @ a_dn
    ops["cdag_d"] = (1, +1, reduce_site_operator([n_dn @ a_up.T, n_up @ a_dn.T], 1))
    ops["cdagF_d"] = (1, +1, reduce_site_operator([n_dn @ a_up.T @ F, n_up @ a_dn.T @ F], 1))
    ops["c_d"] = (1, -1, reduce_site_operator([-n_up @ a_dn, n_dn @ a_up], 1))
    ops["Fc_d"] = (1, -1, reduce_site_operator([-F @ n_up @ a_dn, F @ n_dn @ a_up], 1))
    return ops


def expand_site_operator(name: str):
    """inverse of reduce_site_operator: list of 4x4 component matrices."""
    k, dN, red = site_operators()[name]
    comps = [np.zeros((4, 4)) for _ in range(k + 1)]
    for so, (No, jo) in enumerate(SITE_MULT):
        for si, (Ni, ji) in enumerate(SITE_MULT):
            if red[so, si] == 0.0:
                continue
            C = cg_tensor(ji, k, jo)
            for q in range(k + 1):
                comps[q][SITE_OFFSET[so]:SITE_OFFSET[so] + jo + 1,
                         SITE_OFFSET[si]:SITE_OFFSET[si] + ji + 1] += red[so, si] * C[:, q, :].T
    return comps


# ----------------------------------------------------------------------------------------
# Structure tensors (one per object type on the path)
# ----------------------------------------------------------------------------------------
# Every symmetric tensor = sum over label tuples of  reduced block (x) structure tensor.
# "tilde" normalisation: reduced data of centre / right-type objects is scaled so that the
# plain Frobenius norm of the reduced data equals the norm of the full tensor:
#   left-type  A[a,s,c]      : S = CG(ja js | jc)                         (isometry <=> plain)
#   right-type B[c,s,b]      : S = CG(jc js | jb) * sqrt((jc+1)/(jb+1))   (doubled spins: dim = j+1)
#   centre two-site theta    : S = CG(ja js1|jc) CG(jc js2|jb) / sqrt(jb+1)
#   left env  L[a',w,a]      : S = CG(ja kw | ja')
#   right env R[b',w,b]      : S = CG(jb kw | jb') * sqrt((jb'+1)/(jb+1))  (fixed by the recursion,
#                              verified in tests: identity level <=> R = 1)
#   MPO site  W[wl,s',s,wr]  : S = sum_q CG(kl kop|kr)[ml,q,mr] CG(js kop|js')[ms,q,ms']

def S_left(ja, js, jc):
    return cg_tensor(ja, js, jc)                                   # [ma, ms, mc]


def S_right(jc, js, jb):
    return cg_tensor(jc, js, jb) * sqrt((jc + 1) / (jb + 1))      # [mc, ms, mb]


def S_theta(ja, js1, jc, js2, jb):
    t = np.einsum("asc,ctb->astb", cg_tensor(ja, js1, jc), cg_tensor(jc, js2, jb))
    return t / sqrt(jb + 1)                                        # [ma, m1, m2, mb]


def S_L(jap, kw, ja):
    return np.einsum("awp->pwa", cg_tensor(ja, kw, jap))           # [ma', mw, ma]


def S_R(jbp, kw, jb):
    return np.einsum("bwp->pwb", cg_tensor(jb, kw, jbp)) * sqrt((jbp + 1) / (jb + 1))  # [mb', mw, mb]


def S_W(kl, kop, kr, jsp, js):
    # [ml, ms', ms, mr]
    return np.einsum("lqr,sqp->lpsr", cg_tensor(kl, kop, kr), cg_tensor(js, kop, jsp))


def _project(T, S):
    """coefficient c with T = c * S; asserts T is proportional to S."""
    nrm = float(np.sum(S * S))
    c = float(np.sum(T * S) / nrm)
    assert np.allclose(T, c * S, atol=1e-12), "contraction left the invariant subspace"
    return c


@lru_cache(maxsize=None)
def coef_left_env(jbp, k, jb, jsp, js, kop, jap, kp, ja):
    """L'[a',w',a] += coef * A[b',s',a']^+ L[b',w,b] W[w,s',s,w'] A[b,s,a]   (A left-type)."""
    T = np.einsum("xpy,xwb,wpsv,bsa->yva",
                  S_left(jbp, jsp, jap), S_L(jbp, k, jb), S_W(k, kop, kp, jsp, js),
                  S_left(jb, js, ja), optimize=True)
    return _project(T, S_L(jap, kp, ja))


@lru_cache(maxsize=None)
def coef_right_env(jcp, k, jc, jsp, js, kop, jbp, kp, jb):
    """R[c',w,c] += coef * B[c',s',b']^* W[w,s',s,w'] R[b',w',b] B[c,s,b]     (B right-type)."""
    T = np.einsum("xpy,wpsv,yvb,csb->xwc",
                  S_right(jcp, jsp, jbp), S_W(k, kop, kp, jsp, js), S_R(jbp, kp, jb),
                  S_right(jc, js, jb), optimize=True)
    return _project(T, S_R(jcp, k, jc))


@lru_cache(maxsize=None)
def coef_apply(ja, jap, k, js1, js1p, kop1, km, jc, jcp, js2, js2p, kop2, kp, jb, jbp):
    """y[beta'] += coef * L[a',w,a] theta[beta] R[b',w',b]^T for the path w -(op1)-> wm -(op2)-> w'."""
    T = np.einsum("xwa,wpsu,uqtv,yvb,astb->xpqy",
                  S_L(jap, k, ja), S_W(k, kop1, km, js1p, js1), S_W(km, kop2, kp, js2p, js2),
                  S_R(jbp, kp, jb), S_theta(ja, js1, jc, js2, jb), optimize=True)
    S = S_theta(jap, js1p, jcp, js2p, jbp)
    return float(np.sum(T * S) / np.sum(S * S))       # projection on one tree of the output
