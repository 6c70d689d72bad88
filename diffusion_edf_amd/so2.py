"""Edge-aligned-frame ("SO(2)") form of the depth-wise tensor products  features (x) SH(edge)  (Passaro & Zitnick 2023, eSCN),
generated in float64 with numpy only; `gen_tables.py` emits what the HIP edge kernel needs from here.

The reference evaluates, per edge and path p = (l1, l2, l3) (equiformer/tensor_product_rescale.py:352-382 through e3nn's
`o3.TensorProduct`, graph_attention.py:231-247):

    out[u, k] = w[p, u] * sqrt(2 l3 + 1) * sum_ij C^{l1 l2 l3}_{ijk} x[u, i] Y^{l2}_j(r)          r = unit edge vector

Let g be a rotation with g r = y (e3nn's polar axis).  By equivariance  TP(x, Y(r)) = D(g)^T TP(D(g) x, Y(y)),  and
Y^{l}(y) = sqrt(2 l + 1) e_{m = 0}, so in the rotated ("edge") frame only the m2 = 0 column of the 3j symbol survives:

    out'[u, k] = w[p, u] * sum_i c^p_{ik} x'[u, i],      c^p_{ik} = sqrt(2 l3 + 1) sqrt(2 l2 + 1) C_{i, m2 = 0, k}

and c^p_{ik} != 0 only for |m_i| = |m_k|:  i = k  (l1 + l2 + l3 even),  i = -k with c(-m -> +m) = -c(+m -> -m)  (odd).  Everything that
follows the first depth-wise TP in the attention block -- LinearRS (acts on the channel index), the Gate (scalars x irreps), the second
depth-wise TP with the same SH, its LinearRS -- commutes with D(g), so the whole per-edge chain runs in the edge frame between ONE
rotate-in of the source message and ONE rotate-out of the value.

The rotation is applied like the reference's own Wigner-D recipe (wigner.py:44-81, `D = X(a) J X(b) J X(c)`):
    g = R_x(beta) R_y(gamma),   D^l(g) = J_l X_l(beta) J_l X_l(gamma)
with  cos(gamma) = z / rho, sin(gamma) = -x / rho, cos(beta) = y, sin(beta) = -rho  (rho = sqrt(x^2 + z^2); gamma = 0 when rho = 0):
X_l(angle) mixes the (-m, +m) pairs with cos / sin(m angle) (2 multiply-adds per component), J_l is a constant sparse matrix.  Only
cos / sin of m gamma and m beta are live per edge instead of the (2l+1)^2 entries of D^l.
"""
from __future__ import annotations

import math
from functools import lru_cache

import numpy as np

from . import so3


# ---- frame ------------------------------------------------------------------------------------------------------------------------

def frame_angles(r: np.ndarray):
    """unit vectors r (..., 3) -> (cos gamma, sin gamma, cos beta, sin beta) of the rotation g = R_x(beta) R_y(gamma) with g r = y"""
    r = np.asarray(r, dtype=np.float64)
    x, y, z = r[..., 0], r[..., 1], r[..., 2]
    rho = np.sqrt(x * x + z * z)
    safe = rho > 0
    inv = np.where(safe, 1.0 / np.where(safe, rho, 1.0), 0.0)
    cg = np.where(safe, z * inv, 1.0)
    sg = np.where(safe, -x * inv, 0.0)
    return cg, sg, y, -rho


def rot_in_matrix(l: int, r: np.ndarray) -> np.ndarray:
    """D^l(g) for the edge direction(s) r: (..., 2l+1, 2l+1)"""
    cg, sg, cb, sb = frame_angles(r)
    gamma, beta = np.arctan2(sg, cg), np.arctan2(sb, cb)
    return so3.wigner_D(l, np.zeros_like(gamma), beta, gamma)


# ---- straight-line rotation programs ---------------------------------------------------------------------------------------------
# A program is a list of stages applied to a (2l+1)-vector:
#   ('X', which)   which in {'g', 'b'}: v_i <- cos((l - i) a) v_i + sin((l - i) a) v_{2l - i}      (reference wigner.py:21-42)
#   ('J',)         v <- J_l v
# rot_in = [X(g), J, X(b), J];  rot_out (the transpose) = [J, X(-b), J, X(-g)]

def rot_in_program(l: int):
    return [] if l == 0 else [('X', 'g', +1), ('J',), ('X', 'b', +1), ('J',)]


def rot_out_program(l: int):
    return [] if l == 0 else [('J',), ('X', 'b', -1), ('J',), ('X', 'g', -1)]


def run_program(l: int, prog, v: np.ndarray, r: np.ndarray) -> np.ndarray:
    """apply a program to v (..., 2l+1) for edge direction(s) r (..., 3) -- the float64 model of the generated device code"""
    cg, sg, cb, sb = frame_angles(r)
    ang = {'g': np.arctan2(sg, cg), 'b': np.arctan2(sb, cb)}
    v = np.array(v, dtype=np.float64, copy=True)
    J = so3.J_matrix(l)
    for st in prog:
        if st[0] == 'J':
            v = np.einsum('ij,...j->...i', J, v)
        else:
            a = ang[st[1]] * st[2]
            out = v.copy()
            for i in range(2 * l + 1):
                f = l - i
                if f != 0:
                    out[..., i] = np.cos(f * a) * v[..., i] + np.sin(f * a) * v[..., 2 * l - i]
            v = out
    return v


# ---- edge-frame coefficients -----------------------------------------------------------------------------------------------------

@lru_cache(maxsize=None)
def so2_coeff(l1: int, l2: int, l3: int) -> np.ndarray:
    """c[i, k] = sqrt(2 l3 + 1) sqrt(2 l2 + 1) w3j[i, m2 = 0, k]: the depth-wise TP of the path in the edge frame (see module docstring)"""
    C = so3.wigner_3j(l1, l2, l3) * math.sqrt(2 * l3 + 1) * math.sqrt(2 * l2 + 1)
    c = C[:, l2, :].copy()
    c[np.abs(c) < 1e-13] = 0.0
    return c


def so2_terms(l1: int, l2: int, l3: int):
    """[(k, i, c)] for every output component k the path reaches: out'[k] = c x'[i]  (exactly one source component per k)"""
    c = so2_coeff(l1, l2, l3)
    out = []
    for k in range(2 * l3 + 1):
        nz = [i for i in range(2 * l1 + 1) if c[i, k] != 0.0]
        assert len(nz) <= 1, (l1, l2, l3, k, nz)
        if nz:
            i = nz[0]
            assert abs(i - l1) == abs(k - l3)
            out.append((k, i, float(c[i, k])))
    return out


def so2_ref(l1: int, l2: int, l3: int) -> float:
    """The constant folded into the path's linear-layer weights on the host: the coefficient of its first term (|m| smallest).  What is
    left per term in the kernel is the ratio c / ref (a compile-time constant, +-1 for most paths)."""
    t = so2_terms(l1, l2, l3)
    t = sorted(t, key=lambda e: (abs(e[0] - l3), e[0]))
    return t[0][2]


def dtp_edge_frame(l1: int, l2: int, l3: int, xp: np.ndarray, cns: float = 1.0) -> np.ndarray:
    """out'[..., k] for x' (..., 2l1+1) in the edge frame; cns = the non-scalar SH cut-off factor (multiplies Y_{l2 > 0})"""
    out = np.zeros(xp.shape[:-1] + (2 * l3 + 1,))
    for k, i, c in so2_terms(l1, l2, l3):
        out[..., k] = c * xp[..., i] * (cns if l2 > 0 else 1.0)
    return out


def dtp_direct(l1: int, l2: int, l3: int, x: np.ndarray, r: np.ndarray, cns: float = 1.0) -> np.ndarray:
    """the reference form: sqrt(2 l3 + 1) sum_ij C_ijk x_i Y_j(r), Y_{l2 > 0} multiplied by cns"""
    C = so3.wigner_3j(l1, l2, l3) * math.sqrt(2 * l3 + 1)
    Y = so3.spherical_harmonics(l2, r) * (cns if l2 > 0 else 1.0)
    return np.einsum('ijk,...i,...j->...k', C, x, Y)
