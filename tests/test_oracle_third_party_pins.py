"""Pins of ``oracle/e3ref.py`` against INDEPENDENT third-party implementations present in this image.

The reference's arithmetic lives in e3nn 0.4.4, which cannot be imported here, and the reference ships no tests or
vectors — so the oracle stays "parity unpinned" at the e3nn boundary.  What this image does hold are
sympy and scipy; these tests tie every piece of the oracle's angular algebra to them:

* ``_su2_cg_entry``  ==  ``sympy.physics.quantum.cg.CG``            (complex SU(2) Clebsch-Gordan, all 392 entries)
* oracle real SH     ==  ``scipy.special.sph_harm_y`` through a FITTED unitary change of basis (polar axis y, the
                         component normalisation sqrt(4 pi), the (-i)^l phase, the m = -l..l index order)
* ``wigner_3j``      ==  the real tensor rebuilt from sympy's CG and that fitted basis (every (l1, l2, l3) the kernels
                         support, odd l1+l2+l3 included: the sign of every path relative to the SH basis)
* ``wigner_3j``      ∝   ``sympy.physics.wigner.real_gaunt`` (integrals of three real SH) — for even l1+l2+l3, one
                         constant per triple, of the closed-form magnitude

What remains recollection-only after these: that e3nn 0.4.4's *stored* 3j table carries the same overall
sign per (l1, l2, l3) as this construction, and its ``Irreps.sort`` tie order — both absorbed by random weights.
"""
import itertools
import math

import numpy as np
import pytest
import torch

from oracle import e3ref

TRIPLES = [(l1, l2, l3) for l1 in range(4) for l2 in range(3) for l3 in range(4) if abs(l1 - l2) <= l3 <= l1 + l2]


def _unit_vectors(n, seed):
    rng = np.random.default_rng(seed)
    v = rng.normal(size=(n, 3))
    return v / np.linalg.norm(v, axis=1, keepdims=True)


def _angles(v):
    """e3nn's polar axis is y; its (x, y, z) are the standard frame's (y, z, x)."""
    x, y, z = v.T
    return np.arccos(np.clip(y, -1.0, 1.0)), np.arctan2(x, z)      # polar, azimuth


def test_su2_clebsch_gordan_equals_sympy():
    from sympy import S
    from sympy.physics.quantum.cg import CG

    checked = 0
    for j1 in range(4):
        for j2 in range(3):
            for j3 in range(abs(j1 - j2), j1 + j2 + 1):
                for m1 in range(-j1, j1 + 1):
                    for m2 in range(-j2, j2 + 1):
                        m3 = m1 + m2
                        if abs(m3) > j3:
                            continue
                        want = float(CG(S(j1), S(m1), S(j2), S(m2), S(j3), S(m3)).doit())
                        got = e3ref._su2_cg_entry(j1, m1, j2, m2, j3, m3)
                        assert abs(got - want) < 1e-12, (j1, m1, j2, m2, j3, m3)
                        checked += 1
    assert checked >= 392


def _complex_sh(l, v):
    from scipy import special

    theta, phi = _angles(v)
    if hasattr(special, "sph_harm_y"):
        return np.stack([special.sph_harm_y(l, m, theta, phi) for m in range(-l, l + 1)], axis=1)
    return np.stack([special.sph_harm(m, l, phi, theta) for m in range(-l, l + 1)], axis=1)


def _fitted_basis(l):
    """U_l with  Y_real(oracle) = sqrt(4 pi) (-i)^l  Y_complex(scipy) @ U_l, by least squares over sample directions.
    Nothing of the oracle's own change-of-basis code enters."""
    v = _unit_vectors(200, 10 + l)
    yc = _complex_sh(l, v) * math.sqrt(4.0 * math.pi) * (-1j) ** l
    yr = e3ref.spherical_harmonics([l], torch.tensor(v), normalize=False).numpy()
    u, res, rank, _ = np.linalg.lstsq(yc, yr.astype(complex), rcond=None)
    assert rank == 2 * l + 1
    assert np.abs(yc @ u - yr).max() < 1e-12            # the real SH span exactly scipy's degree-l harmonics
    return u


@pytest.mark.parametrize("l", [0, 1, 2, 3])
def test_real_sh_is_a_unitary_image_of_scipy_complex_sh(l):
    u = _fitted_basis(l)
    assert np.abs(u.conj().T @ u - np.eye(2 * l + 1)).max() < 1e-12      # unitary: component normalisation sqrt(4 pi)
    # the fitted basis is the conjugate of the oracle's real->complex matrix (a check of _q_real_to_complex, not an input)
    assert np.abs(u - np.conj(e3ref._q_real_to_complex(l).numpy())).max() < 1e-12
    # index l (m = 0) is the zonal harmonic about y: P_l(y) sqrt(2l+1)
    v = _unit_vectors(20, 3)
    yr = e3ref.spherical_harmonics([l], torch.tensor(v), normalize=False).numpy()
    from scipy.special import eval_legendre

    assert np.abs(yr[:, l] - math.sqrt(2 * l + 1) * eval_legendre(l, v[:, 1])).max() < 1e-12


@pytest.mark.parametrize("l1,l2,l3", TRIPLES)
def test_real_wigner_3j_rebuilt_from_sympy_cg_and_scipy_basis(l1, l2, l3):
    """C_real[j, l, m] = sum U1[i, j] U2[k, l] conj(U3)[n, m] <l1 i; l2 k | l3 n> / norm, with U_l the bases fitted
    against scipy above (they carry the (-i)^l phases that make the result real) and the CG coefficients from sympy."""
    from sympy import S
    from sympy.physics.quantum.cg import CG

    c = np.zeros((2 * l1 + 1, 2 * l2 + 1, 2 * l3 + 1), dtype=complex)
    for m1 in range(-l1, l1 + 1):
        for m2 in range(-l2, l2 + 1):
            if abs(m1 + m2) <= l3:
                c[l1 + m1, l2 + m2, l3 + m1 + m2] = float(CG(S(l1), S(m1), S(l2), S(m2), S(l3), S(m1 + m2)).doit())
    # the fitted U is conj(Q) with Q = (-i)^l q: the oracle's formula is einsum(Q1, Q2, conj(Q3^T), C)
    q1, q2, q3 = (np.conj(_fitted_basis(l)) for l in (l1, l2, l3))
    real = np.einsum("ij,kl,mn,ikn->jlm", q1, q2, np.conj(q3.T), c)
    assert np.abs(real.imag).max() < 1e-12
    real = real.real / np.linalg.norm(real.real)
    got = e3ref.wigner_3j(l1, l2, l3).numpy()
    assert np.abs(got - real).max() < 1e-12


@pytest.mark.parametrize("l1,l2,l3", [t for t in TRIPLES if sum(t) % 2 == 0])
def test_real_wigner_3j_is_proportional_to_sympy_real_gaunt(l1, l2, l3):
    """For even l1+l2+l3 the integral of three real harmonics is an invariant tensor, hence a multiple of the real 3j:
    one constant per triple, zero patterns identical, with NO per-component sign fix-up — sympy's real harmonics
    (cos-type for m > 0, sin-type for m < 0, no Condon-Shortley factor left over) are the oracle's basis with the polar
    axis renamed.  (Odd sums integrate to zero; the test above covers them.)"""
    from sympy.physics.wigner import real_gaunt

    got = e3ref.wigner_3j(l1, l2, l3).numpy()
    gaunt = np.zeros_like(got)
    for (i, j, k) in itertools.product(range(2 * l1 + 1), range(2 * l2 + 1), range(2 * l3 + 1)):
        g = float(real_gaunt(l1, l2, l3, i - l1, j - l2, k - l3))
        gaunt[i, j, k] = g
    assert np.linalg.norm(gaunt) > 0
    const = float((gaunt * got).sum() / (got * got).sum())
    assert np.abs(gaunt - const * got).max() < 1e-12 * max(1.0, abs(const))
    # closed form of the constant: sqrt((2l1+1)(2l2+1)(2l3+1) / 4 pi) (l1 l2 l3; 0 0 0), up to the construction's sign
    from sympy.physics.wigner import wigner_3j as w3j

    mag = math.sqrt((2 * l1 + 1) * (2 * l2 + 1) * (2 * l3 + 1) / (4.0 * math.pi)) * abs(float(w3j(l1, l2, l3, 0, 0, 0)))
    assert abs(abs(const) - mag) < 1e-10
