"""SO(3) constants for the score-head hot path, generated in float64 with numpy only.

Everything the HIP kernels need about irreps lives here and is emitted as a C header by
``diffusion_edf_amd/gen_tables.py``:

* real spherical harmonics, 'component' normalisation, y = polar axis
  (what ``o3.SphericalHarmonics(normalize=True, normalization='component')`` evaluates at
  reference ``diffusion_edf/graph_parser.py:135``),
* the Wigner ``J_l`` matrices used by reference ``diffusion_edf/wigner.py:44-81``
  (``D^l = X(a) J X(b) J X(c)``; the reference loads them from e3nn / an LFS stub, ``w3j.py:6-10``),
* real-basis Wigner 3j symbols (``o3.wigner_3j``) that e3nn's ``o3.TensorProduct`` contracts
  (reference ``equiformer/tensor_product_rescale.py:38-42``),
* the ``normalize2mom`` constants of ``equiformer/fast_activation.py:69``.

e3nn 0.4.4 is not vendored in the reference; the constructions below restate its published algorithm
(Racah formula + real/complex change of basis).  They are pinned by invariants in ``tests/test_so3.py``
(equivariance under the reference's own Wigner-D recipe) and cross-checked against the independent
derivation in ``oracle/so3_oracle.py``.
"""
from __future__ import annotations

import math
from fractions import Fraction
from functools import lru_cache

import numpy as np

# --------------------------------------------------------------------------------------------------
# spherical harmonics (l <= 3)
# --------------------------------------------------------------------------------------------------


def spherical_harmonics(l: int, v: np.ndarray, normalize: bool = True) -> np.ndarray:
    """Real SH of degree ``l`` of vectors ``v`` (..., 3); component normalisation (sum_m Y_m^2 = 2l+1
    on the unit sphere).  Zero vectors map to Y_0 = 1, Y_{l>0} = 0 (F.normalize semantics)."""
    v = np.asarray(v, dtype=np.float64)
    if normalize:
        n = np.linalg.norm(v, axis=-1, keepdims=True)
        v = v / np.maximum(n, 1e-12)
    x, y, z = v[..., 0], v[..., 1], v[..., 2]
    if l == 0:
        return np.ones(v.shape[:-1] + (1,))
    if l == 1:
        return math.sqrt(3.0) * np.stack([x, y, z], axis=-1)
    a = x * z
    b = (z * z - x * x) / 2
    rho = x * x + z * z
    if l == 2:
        s3 = math.sqrt(3.0)
        return math.sqrt(5.0) * np.stack([s3 * a, s3 * x * y, y * y - rho / 2, s3 * y * z, s3 * b], axis=-1)
    if l == 3:
        return np.stack([
            math.sqrt(35.0 / 2) * (a * z + b * x),
            math.sqrt(105.0) * a * y,
            math.sqrt(21.0 / 8) * (4 * y * y - rho) * x,
            math.sqrt(7.0) / 2 * y * (2 * y * y - 3 * rho),
            math.sqrt(21.0 / 8) * z * (4 * y * y - rho),
            math.sqrt(105.0) * b * y,
            math.sqrt(35.0 / 2) * (b * z - a * x),
        ], axis=-1)
    raise NotImplementedError(l)


# --------------------------------------------------------------------------------------------------
# J matrices and Wigner D (reference recipe, wigner.py:21-81)
# --------------------------------------------------------------------------------------------------

_J1 = np.array([[0.0, 1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, -1.0]])


@lru_cache(maxsize=None)
def J_matrix(l: int) -> np.ndarray:
    """``J_l`` = D^l of the rotation by pi about (x+y)/sqrt2 (x<->y, z->-z).  Solved by least squares
    from ``Y_l(J_1 p) = J_l Y_l(p)`` so that no e3nn constant file is needed."""
    if l == 0:
        return np.ones((1, 1))
    rng = np.random.default_rng(1234 + l)
    p = rng.normal(size=(64 * (2 * l + 1), 3))
    p /= np.linalg.norm(p, axis=-1, keepdims=True)
    A = spherical_harmonics(l, p)                 # (n, 2l+1)
    B = spherical_harmonics(l, p @ _J1.T)         # Y(J1 p)
    # B = A @ J^T
    Jt, *_ = np.linalg.lstsq(A, B, rcond=None)
    J = Jt.T
    J[np.abs(J) < 1e-12] = 0.0
    return J


def z_rot_mat(angle: np.ndarray, l: int) -> np.ndarray:
    """Restates reference wigner.py:21-42 (`_z_rot_mat`, in fact a rotation about y)."""
    angle = np.atleast_1d(np.asarray(angle, dtype=np.float64))
    M = np.zeros(angle.shape + (2 * l + 1, 2 * l + 1))
    inds = np.arange(2 * l + 1)
    rev = np.arange(2 * l, -1, -1)
    freq = np.arange(l, -l - 1, -1, dtype=np.float64)
    M[..., inds, rev] = np.sin(freq * angle[..., None])
    M[..., inds, inds] = np.cos(freq * angle[..., None])
    return M


def wigner_D(l: int, alpha, beta, gamma) -> np.ndarray:
    """``X(alpha) J X(beta) J X(gamma)`` — reference wigner.py:76-81."""
    J = J_matrix(l)
    return z_rot_mat(alpha, l) @ J @ z_rot_mat(beta, l) @ J @ z_rot_mat(gamma, l)


# --------------------------------------------------------------------------------------------------
# Wigner 3j in the real basis (e3nn `o3.wigner_3j` construction)
# --------------------------------------------------------------------------------------------------


def _fact(n: int) -> int:
    return math.factorial(n)


def _su2_cg_coeff(j1, m1, j2, m2, j3, m3) -> float:
    """<j1 m1 j2 m2 | j3 m3> by the Racah formula (exact rational arithmetic under the root)."""
    if m3 != m1 + m2:
        return 0.0
    vmin = int(max(-j1 + j2 + m3, -j1 + m1, 0))
    vmax = int(min(j2 + j3 + m1, j3 - j1 + j2, j3 + m3))
    C = Fraction((2 * j3 + 1) * _fact(j3 + j1 - j2) * _fact(j3 - j1 + j2) * _fact(j1 + j2 - j3)
                 * _fact(j3 + m3) * _fact(j3 - m3),
                 _fact(j1 + j2 + j3 + 1) * _fact(j1 - m1) * _fact(j1 + m1) * _fact(j2 - m2) * _fact(j2 + m2))
    S = Fraction(0)
    for v in range(vmin, vmax + 1):
        S += Fraction((-1) ** (v + j2 + m2) * _fact(j2 + j3 + m1 - v) * _fact(j1 - m1 + v),
                      _fact(v) * _fact(j3 - j1 + j2 - v) * _fact(j3 + m3 - v) * _fact(v + j1 - j2 - m3))
    return math.sqrt(float(C)) * float(S)


def _su2_cg(j1: int, j2: int, j3: int) -> np.ndarray:
    mat = np.zeros((2 * j1 + 1, 2 * j2 + 1, 2 * j3 + 1))
    for m1 in range(-j1, j1 + 1):
        for m2 in range(-j2, j2 + 1):
            m3 = m1 + m2
            if abs(m3) <= j3:
                mat[j1 + m1, j2 + m2, j3 + m3] = _su2_cg_coeff(j1, m1, j2, m2, j3, m3)
    return mat


def _real_to_complex(l: int) -> np.ndarray:
    q = np.zeros((2 * l + 1, 2 * l + 1), dtype=np.complex128)
    s2 = 1 / math.sqrt(2)
    for m in range(-l, 0):
        q[l + m, l + abs(m)] = s2
        q[l + m, l - abs(m)] = -1j * s2
    q[l, l] = 1
    for m in range(1, l + 1):
        q[l + m, l + abs(m)] = (-1) ** m * s2
        q[l + m, l - abs(m)] = 1j * (-1) ** m * s2
    return (-1j) ** l * q


@lru_cache(maxsize=None)
def wigner_3j(l1: int, l2: int, l3: int) -> np.ndarray:
    """Real-basis 3j symbol, Frobenius norm 1, shape (2l1+1, 2l2+1, 2l3+1)."""
    assert abs(l1 - l2) <= l3 <= l1 + l2
    Q1, Q2, Q3 = _real_to_complex(l1), _real_to_complex(l2), _real_to_complex(l3)
    C = _su2_cg(l1, l2, l3).astype(np.complex128)
    C = np.einsum('ij,kl,mn,ikn->jlm', Q1, Q2, np.conj(Q3.T), C)
    assert np.abs(C.imag).max() < 1e-12, (l1, l2, l3)
    C = C.real.copy()
    C /= np.linalg.norm(C)
    C[np.abs(C) < 1e-14] = 0.0
    return C


# --------------------------------------------------------------------------------------------------
# normalize2mom constants (e3nn.math.normalize2mom: 1e6 float64 N(0,1) samples, torch CPU seed 0)
# --------------------------------------------------------------------------------------------------

# Values computed with this container's torch following that recipe; tests/test_so3.py::test_normalize2mom_constants re-derives
# them (the provenance check).  Hard-coded because the product must not depend on a Monte-Carlo run at import.
NORM2MOM_SILU = 1.6791767923989418
NORM2MOM_SIGMOID = 1.8467055342154763
NORM2MOM_SLRELU02 = 1.531320475574866


# --------------------------------------------------------------------------------------------------
# irreps helpers
# --------------------------------------------------------------------------------------------------


def parse_irreps(s) -> list[tuple[int, int]]:
    """'64x0e+32x1e' -> [(64,0),(32,1)].  Only even parity is supported, as in the reference
    (wigner.py:237-239 raises for p != 1)."""
    if isinstance(s, (list, tuple)):
        return [(int(m), int(l)) for m, l in s]
    out = []
    for tok in str(s).replace(' ', '').split('+'):
        if not tok:
            continue
        if 'x' in tok:
            mul, ir = tok.split('x')
        else:
            mul, ir = '1', tok
        if ir[-1] != 'e':
            raise NotImplementedError(f"odd parity irreps are not supported: {s}")
        out.append((int(mul), int(ir[:-1])))
    return out


def irreps_dim(irreps) -> int:
    return sum(m * (2 * l + 1) for m, l in parse_irreps(irreps))


def irreps_str(irreps) -> str:
    return '+'.join(f"{m}x{l}e" for m, l in parse_irreps(irreps))
