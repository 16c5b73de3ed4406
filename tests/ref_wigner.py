"""Closed-form SU(2) recoupling coefficients used by the planner (doubled spins throughout).

These replace, on the product side, what TensorKit obtains from F-symbols / fusion-tree
transformers.  tests/test_wigner.py checks every
function against brute-force contraction of Clebsch-Gordan tensors in oracle/su2.py.
"""
from __future__ import annotations

from functools import lru_cache
from math import factorial, sqrt


def triangle(a: int, b: int, c: int) -> bool:
    return abs(a - b) <= c <= a + b and (a + b + c) % 2 == 0


def _delta(a: int, b: int, c: int) -> float:
    return (factorial((a + b - c) // 2) * factorial((a - b + c) // 2) * factorial((-a + b + c) // 2)
            / factorial((a + b + c) // 2 + 1))


@lru_cache(maxsize=None)
def wigner6j(j1: int, j2: int, j3: int, j4: int, j5: int, j6: int) -> float:
    """{j1 j2 j3; j4 j5 j6}, Racah formula."""
    if not (triangle(j1, j2, j3) and triangle(j1, j5, j6) and triangle(j4, j2, j6) and triangle(j4, j5, j3)):
        return 0.0
    pref = sqrt(_delta(j1, j2, j3) * _delta(j1, j5, j6) * _delta(j4, j2, j6) * _delta(j4, j5, j3))
    a1, a2, a3, a4 = (j1 + j2 + j3) // 2, (j1 + j5 + j6) // 2, (j4 + j2 + j6) // 2, (j4 + j5 + j3) // 2
    b1, b2, b3 = (j1 + j2 + j4 + j5) // 2, (j2 + j3 + j5 + j6) // 2, (j3 + j1 + j6 + j4) // 2
    s = 0.0
    for t in range(max(a1, a2, a3, a4), min(b1, b2, b3) + 1):
        s += ((-1) ** t * factorial(t + 1)
              / (factorial(t - a1) * factorial(t - a2) * factorial(t - a3) * factorial(t - a4)
                 * factorial(b1 - t) * factorial(b2 - t) * factorial(b3 - t)))
    return pref * s


@lru_cache(maxsize=None)
def wigner9j(j1, j2, j3, j4, j5, j6, j7, j8, j9) -> float:
    """{j1 j2 j3; j4 j5 j6; j7 j8 j9} as a sum over products of three 6j symbols."""
    if not (triangle(j1, j2, j3) and triangle(j4, j5, j6) and triangle(j7, j8, j9)
            and triangle(j1, j4, j7) and triangle(j2, j5, j8) and triangle(j3, j6, j9)):
        return 0.0
    lo = max(abs(j1 - j9), abs(j4 - j8), abs(j2 - j6))
    hi = min(j1 + j9, j4 + j8, j2 + j6)
    s = 0.0
    for x in range(lo, hi + 1, 2):
        s += ((-1) ** x * (x + 1) * wigner6j(j1, j4, j7, j8, j9, x) * wigner6j(j2, j5, j8, j4, x, j6)
              * wigner6j(j3, j6, j9, x, j1, j2))
    return s
