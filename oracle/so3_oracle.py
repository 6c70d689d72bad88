"""ORACLE (test infrastructure, never shipped in the product path).

Independent derivation of the SO(3) constants used by ``oracle/restatement.py``:

* Wigner-D by the reference's recipe ``D = X(a) J X(b) J X(c)`` (reference diffusion_edf/wigner.py:21-81),
  with ``J_l`` obtained as the representation matrix of the x<->y, z->-z rotation,
* real spherical harmonics by *recursion* ``Y_{l+1} ~ C(l,1,l+1) . (Y_l (x) Y_1)`` (not the closed forms
  the product uses), component-normalised, y polar,
* real Wigner-3j symbols as the *null space* of the invariance condition
  ``(D1 (x) D2 (x) D3) C = C`` (not the Racah formula the product uses).  The null space fixes C up to one
  sign per (l1,l2,l3) block; that bit is anchored by ``_SIGN_ANCHOR`` (sign of the first non-zero entry in
  row-major order under the e3nn construction: Racah CG + real/complex change of basis).

e3nn 0.4.4 (pinned by reference setup.py:28) is not vendored in /root/reference and not installable here,
so the convention itself is "parity unpinned" with respect to e3nn's shipped constants; what *is* pinned
(tests/test_so3.py) is that both derivations agree and are equivariant under the reference's Wigner-D.
"""
from __future__ import annotations

import math
from functools import lru_cache

import numpy as np

_J1 = np.array([[0.0, 1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, -1.0]])


def _z_rot(angle: float, l: int) -> np.ndarray:
    # reference wigner.py:21-42
    M = np.zeros((2 * l + 1, 2 * l + 1))
    for a in range(2 * l + 1):
        f = l - a
        M[a, 2 * l - a] = math.sin(f * angle)
        M[a, a] = math.cos(f * angle)
    return M


def _rot_y(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def _rot_x(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]])


def _sh1(v):
    return math.sqrt(3.0) * v


# sign of the first non-zero entry (row-major) of wigner_3j(l1,l2,l3) in the e3nn construction
_SIGN_ANCHOR = {
    (0, 0, 0): 1, (0, 1, 1): 1, (0, 2, 2): 1, (0, 3, 3): 1, (1, 0, 1): 1, (1, 1, 0): 1, (1, 1, 1): 1,
    (1, 1, 2): -1, (1, 2, 1): 1, (1, 2, 2): -1, (1, 2, 3): -1, (1, 3, 2): 1, (1, 3, 3): -1, (2, 0, 2): 1,
    (2, 1, 1): 1, (2, 1, 2): 1, (2, 1, 3): -1, (2, 2, 0): 1, (2, 2, 1): -1, (2, 2, 2): -1, (2, 2, 3): 1,
    (2, 3, 1): 1, (2, 3, 2): -1, (2, 3, 3): -1, (3, 0, 3): 1, (3, 1, 2): 1, (3, 1, 3): 1, (3, 2, 1): 1,
    (3, 2, 2): 1, (3, 2, 3): -1, (3, 3, 0): 1, (3, 3, 1): -1, (3, 3, 2): -1, (3, 3, 3): 1,
}


@lru_cache(maxsize=None)
def J(l: int) -> np.ndarray:
    """J_l by recursion on l through the 3j symbols would be circular; instead use the fact that
    J_l = D^l(R_J) and D^l can be read off from how degree-l harmonic polynomials transform.  We build
    the l-th harmonics as symmetric traceless tensors implicitly: fit D from Y_l samples produced by the
    recursion-free *Legendre* form of the real harmonics (see ``sh``)."""
    if l == 0:
        return np.ones((1, 1))
    rng = np.random.default_rng(99 + l)
    p = rng.normal(size=(40 * (2 * l + 1), 3))
    p /= np.linalg.norm(p, axis=1, keepdims=True)
    A = sh(l, p)
    B = sh(l, p @ _J1.T)
    X, *_ = np.linalg.lstsq(A, B, rcond=None)
    Jm = X.T
    Jm[np.abs(Jm) < 1e-12] = 0
    return Jm


def sh(l: int, v: np.ndarray) -> np.ndarray:
    """Real SH via associated Legendre functions in the polar angle measured from **y**, azimuth in the
    (z, x) plane, ordered m = -l..l as [sin(|m| phi) terms ..., m=0, cos(m phi) terms ...], component
    normalised.  (Independent of the closed-form polynomials in diffusion_edf_amd/so3.py.)"""
    v = np.asarray(v, dtype=np.float64)
    n = np.linalg.norm(v, axis=-1, keepdims=True)
    u = v / np.maximum(n, 1e-12)
    x, y, z = u[..., 0], u[..., 1], u[..., 2]
    ct = y
    st = np.sqrt(np.maximum(0.0, 1 - ct * ct))
    phi = np.arctan2(x, z)
    out = np.zeros(u.shape[:-1] + (2 * l + 1,))
    for m in range(0, l + 1):
        # associated Legendre P_l^m(ct) without Condon-Shortley phase
        P = _legendre(l, m, ct, st)
        N = math.sqrt((2 * l + 1) * math.factorial(l - m) / math.factorial(l + m))
        if m == 0:
            out[..., l] = N * P
        else:
            out[..., l + m] = math.sqrt(2) * N * P * np.cos(m * phi)
            out[..., l - m] = math.sqrt(2) * N * P * np.sin(m * phi)
    zero = (n[..., 0] < 1e-12)
    if l > 0:
        out[zero] = 0.0
    return out


def _legendre(l, m, ct, st):
    # P_m^m = (2m-1)!! st^m ; P_{m+1}^m = ct (2m+1) P_m^m ; upward recursion
    pmm = np.ones_like(ct)
    for k in range(1, m + 1):
        pmm = pmm * (2 * k - 1) * st
    if l == m:
        return pmm
    pm1 = ct * (2 * m + 1) * pmm
    if l == m + 1:
        return pm1
    for ll in range(m + 2, l + 1):
        pll = ((2 * ll - 1) * ct * pm1 - (ll + m - 1) * pmm) / (ll - m)
        pmm, pm1 = pm1, pll
    return pm1


def wigner_D_angles(l: int, a: float, b: float, c: float) -> np.ndarray:
    Jl = J(l)
    return _z_rot(a, l) @ Jl @ _z_rot(b, l) @ Jl @ _z_rot(c, l)


@lru_cache(maxsize=None)
def w3j(l1: int, l2: int, l3: int) -> np.ndarray:
    d1, d2, d3 = 2 * l1 + 1, 2 * l2 + 1, 2 * l3 + 1
    n = d1 * d2 * d3
    rng = np.random.default_rng(7)
    B = np.zeros((n, n))
    for _ in range(4):
        a, b, c = rng.uniform(0, 2 * math.pi), rng.uniform(0.3, 2.8), rng.uniform(0, 2 * math.pi)
        D = np.kron(np.kron(wigner_D_angles(l1, a, b, c), wigner_D_angles(l2, a, b, c)), wigner_D_angles(l3, a, b, c))
        M = D - np.eye(n)
        B += M.T @ M
    w, v = np.linalg.eigh(B)
    assert w[0] < 1e-10 and (n == 1 or w[1] > 1e-6), (l1, l2, l3, w[:3])
    C = v[:, 0].reshape(d1, d2, d3)
    C[np.abs(C) < 1e-12] = 0
    C /= np.linalg.norm(C)
    first = C.flatten()[np.nonzero(C.flatten())[0][0]]
    if np.sign(first) != _SIGN_ANCHOR[(l1, l2, l3)]:
        C = -C
    return C


# normalize2mom constants, re-derived in tests from the recalled e3nn recipe (torch CPU generator seed 0)
C_SILU = 1.6791767923989418
C_SIGMOID = 1.8467055342154763
C_SLRELU = 1.531320475574866
