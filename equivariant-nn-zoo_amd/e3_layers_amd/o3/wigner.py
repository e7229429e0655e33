"""Real-basis Wigner 3j symbols for the host-side path tables.

Specification: SURVEY.md appendix A.2 (the operator the reference reaches through
``e3nn.o3.wigner_3j``, call site ``e3_layers/nn/output.py:172``; implicitly used by every
``o3.TensorProduct`` built in ``e3_layers/nn/pointwise.py:78`` and
``e3_layers/nn/message_passing.py:83``).  Shape ``[2l1+1, 2l2+1, 2l3+1]``, Frobenius
norm 1, real basis with the m-order / (x, y, z) convention of e3nn (polar axis y).

The numbers produced here are baked, as literals, into the generated HIP header
``csrc/e3k_cg_gen.h`` by ``tools/gen_cg.py``; the Python copy is used to size tables and in
tests.  Everything is float64 + exact integer factorials.
"""
from __future__ import annotations

from fractions import Fraction
from functools import lru_cache
from math import factorial, sqrt
from typing import List, Tuple

import numpy as np


def _su2_cg_coeff(j1: int, m1: int, j2: int, m2: int, j3: int, m3: int) -> float:
    """<j1 m1 j2 m2 | j3 m3> by Racah's closed form (integer spins only)."""
    if m3 != m1 + m2:
        return 0.0
    vmin = max(-j1 + j2 + m3, -j1 + m1, 0)
    vmax = min(j2 + j3 + m1, j3 - j1 + j2, j3 + m3)
    f = factorial
    pref = Fraction(
        (2 * j3 + 1) * f(j3 + j1 - j2) * f(j3 - j1 + j2) * f(j1 + j2 - j3) * f(j3 + m3) * f(j3 - m3),
        f(j1 + j2 + j3 + 1) * f(j1 - m1) * f(j1 + m1) * f(j2 - m2) * f(j2 + m2),
    )
    acc = Fraction(0)
    for v in range(vmin, vmax + 1):
        term = Fraction(
            f(j2 + j3 + m1 - v) * f(j1 - m1 + v),
            f(v) * f(j3 - j1 + j2 - v) * f(j3 + m3 - v) * f(v + j1 - j2 - m3),
        )
        acc += term if (v + j2 + m2) % 2 == 0 else -term
    return sqrt(pref) * float(acc)


def _su2_cg(j1: int, j2: int, j3: int) -> np.ndarray:
    out = np.zeros((2 * j1 + 1, 2 * j2 + 1, 2 * j3 + 1), dtype=np.float64)
    if not (abs(j1 - j2) <= j3 <= j1 + j2):
        return out
    for m1 in range(-j1, j1 + 1):
        for m2 in range(-j2, j2 + 1):
            m3 = m1 + m2
            if abs(m3) <= j3:
                out[j1 + m1, j2 + m2, j3 + m3] = _su2_cg_coeff(j1, m1, j2, m2, j3, m3)
    return out


def _real_to_complex(l: int) -> np.ndarray:
    """Change of basis Q_l (rows: complex m, columns: real index), times (-i)^l."""
    q = np.zeros((2 * l + 1, 2 * l + 1), dtype=np.complex128)
    s = 1.0 / sqrt(2.0)
    for m in range(-l, 0):
        q[l + m, l + abs(m)] = s
        q[l + m, l - abs(m)] = -1j * s
    q[l, l] = 1.0
    for m in range(1, l + 1):
        sign = -1.0 if m % 2 else 1.0
        q[l + m, l + abs(m)] = sign * s
        q[l + m, l - abs(m)] = 1j * sign * s
    return ((-1j) ** l) * q


@lru_cache(maxsize=None)
def _wigner_3j_cached(l1: int, l2: int, l3: int) -> np.ndarray:
    q1, q2, q3 = _real_to_complex(l1), _real_to_complex(l2), _real_to_complex(l3)
    c = _su2_cg(l1, l2, l3).astype(np.complex128)
    c = np.einsum("ij,kl,mn,ikn->jlm", q1, q2, np.conj(q3.T), c)
    if np.abs(c.imag).max() > 1e-9:
        raise AssertionError("real-basis Clebsch-Gordan tensor is not real")
    c = np.ascontiguousarray(c.real)
    c /= np.linalg.norm(c)
    c[np.abs(c) < 1e-14] = 0.0
    c.setflags(write=False)
    return c


def wigner_3j(l1: int, l2: int, l3: int) -> np.ndarray:
    """Real Wigner 3j tensor C[i, j, k] (read-only float64 array)."""
    l1, l2, l3 = int(l1), int(l2), int(l3)
    if not (abs(l1 - l2) <= l3 <= l1 + l2):
        raise ValueError(f"({l1},{l2},{l3}) violates the triangle rule")
    return _wigner_3j_cached(l1, l2, l3)


def cg_nonzeros(l1: int, l2: int, l3: int) -> List[Tuple[int, int, int, float]]:
    """Sparse list ``(i, j, k, value)`` of the non-zero entries, ordered by (k, i, j)."""
    c = wigner_3j(l1, l2, l3)
    idx = np.argwhere(c != 0.0)
    items = [(int(i), int(j), int(k), float(c[i, j, k])) for i, j, k in idx]
    items.sort(key=lambda t: (t[2], t[0], t[1]))
    return items
