"""closed-form recoupling coefficients of the product planner vs brute-force CG contraction (oracle)"""
import itertools

import numpy as np

import ref_planner as pl
from ref_wigner import triangle, wigner6j, wigner9j
from oracle import su2


def test_6j_orthogonality():
    for a, b, d, e in itertools.product(range(4), repeat=4):
        for c in su2.couple(a, b):
            for cp in su2.couple(a, b):
                s = 0.0
                for f in range(0, 9):
                    s += (f + 1) * wigner6j(a, b, c, d, e, f) * wigner6j(a, b, cp, d, e, f)
                if triangle(d, e, c) and triangle(d, e, cp):
                    assert abs(s - (1.0 / (c + 1) if c == cp else 0.0)) < 1e-12


def _tuples(kmax=2, jmax=4):
    for jb in range(jmax + 1):
        for k in range(kmax + 1):
            for jbp in su2.couple(jb, k):
                for js in (0, 1):
                    for kop in range(kmax + 1):
                        for jsp in su2.couple(js, kop):
                            if jsp > 1:
                                continue
                            for ja in su2.couple(jb, js):
                                for kp in su2.couple(k, kop):
                                    for jap in su2.couple(ja, kp):
                                        if triangle(jbp, jsp, jap):
                                            yield (jbp, k, jb, jsp, js, kop, jap, kp, ja)


def test_env_coefficients_match_brute_force():
    n = 0
    for t in _tuples():
        assert abs(pl.coef_left(*t) - su2.coef_left_env(*t)) < 1e-12
        assert abs(pl.coef_right(*t) - su2.coef_right_env(*t)) < 1e-12
        n += 1
    assert n > 300


def test_apply_coefficient_matches_brute_force():
    rng = np.random.default_rng(0)
    cases = []
    for ja in range(4):
        for k in (0, 1):
            for jap in su2.couple(ja, k):
                for js1, kop1 in itertools.product((0, 1), (0, 1)):
                    for js1p in su2.couple(js1, kop1):
                        if js1p > 1:
                            continue
                        for jc in su2.couple(ja, js1):
                            for km in su2.couple(k, kop1):
                                for jcp in su2.couple(jc, km):
                                    if not triangle(jap, js1p, jcp):
                                        continue
                                    for js2, kop2 in itertools.product((0, 1), (0, 1)):
                                        for js2p in su2.couple(js2, kop2):
                                            if js2p > 1:
                                                continue
                                            for jb in su2.couple(jc, js2):
                                                for kp in su2.couple(km, kop2):
                                                    for jbp in su2.couple(jb, kp):
                                                        if triangle(jcp, js2p, jbp):
                                                            cases.append((ja, jap, k, js1, js1p, kop1, km, jc, jcp,
                                                                          js2, js2p, kop2, kp, jb, jbp))
    sel = rng.choice(len(cases), size=250, replace=False)
    for i in sel:
        c = cases[i]
        assert abs(pl.coef_apply(*c) - su2.coef_apply(*c)) < 1e-12
