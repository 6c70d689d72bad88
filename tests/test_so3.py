"""SO(3) constants: product generator vs the oracle's independent derivation, and the invariants of SURVEY §8(c)."""
import math

import numpy as np
import pytest
import torch

from diffusion_edf_amd import gen_tables, so3
from oracle import so3_oracle as oso3

TRIPLES = [(a, b, c) for a in range(4) for b in range(4) for c in range(abs(a - b), min(3, a + b) + 1)]


def _rand_angles(n, seed=0):
    rng = np.random.default_rng(seed)
    return rng.uniform(0, 2 * np.pi, n), rng.uniform(0.2, 2.9, n), rng.uniform(0, 2 * np.pi, n)


def _rot(a, b, c):
    def ry(t):
        return np.array([[np.cos(t), 0, np.sin(t)], [0, 1, 0], [-np.sin(t), 0, np.cos(t)]])

    def rx(t):
        return np.array([[1, 0, 0], [0, np.cos(t), -np.sin(t)], [0, np.sin(t), np.cos(t)]])
    return ry(a) @ rx(b) @ ry(c)


@pytest.mark.parametrize("l", [0, 1, 2, 3])
def test_sh_two_derivations_agree_and_are_component_normalised(l):
    p = np.random.default_rng(l).normal(size=(100, 3))
    a, b = so3.spherical_harmonics(l, p), oso3.sh(l, p)
    assert np.abs(a - b).max() < 1e-12
    assert np.abs((a ** 2).sum(-1) - (2 * l + 1)).max() < 1e-12
    assert np.abs(so3.spherical_harmonics(l, np.zeros((1, 3)))[0] - (1.0 if l == 0 else 0.0)).max() == 0


@pytest.mark.parametrize("l", [1, 2, 3])
def test_J_and_wigner_D(l):
    J = so3.J_matrix(l)
    assert np.abs(J - oso3.J(l)).max() < 1e-12
    assert np.abs(J - J.T).max() < 1e-12 and np.abs(J @ J - np.eye(2 * l + 1)).max() < 1e-12
    al, be, ga = _rand_angles(4, l)
    p = np.random.default_rng(5).normal(size=(20, 3))
    for a, b, c in zip(al, be, ga):
        D = so3.wigner_D(l, a, b, c)[0]
        assert np.abs(D @ D.T - np.eye(2 * l + 1)).max() < 1e-12
        R = _rot(a, b, c)          # l = 1 irrep basis is (x, y, z):  D^1 = R
        if l == 1:
            assert np.abs(D - R).max() < 1e-12
        assert np.abs(so3.spherical_harmonics(l, p @ R.T) - so3.spherical_harmonics(l, p) @ D.T).max() < 1e-12
    # homomorphism
    D1, D2 = so3.wigner_D(l, al[0], be[0], ga[0])[0], so3.wigner_D(l, al[1], be[1], ga[1])[0]
    R12 = _rot(al[0], be[0], ga[0]) @ _rot(al[1], be[1], ga[1])
    Y = so3.spherical_harmonics(l, p)
    assert np.abs(so3.spherical_harmonics(l, p @ R12.T) - Y @ (D1 @ D2).T).max() < 1e-12


def test_J_known_values():
    assert np.abs(so3.J_matrix(1) - np.array([[0, 1, 0], [1, 0, 0], [0, 0, -1.0]])).max() < 1e-14
    J2 = so3.J_matrix(2)
    assert abs(J2[0, 3] + 1) < 1e-12 and abs(J2[1, 1] - 1) < 1e-12 and abs(J2[2, 2] + 0.5) < 1e-12
    assert abs(J2[2, 4] + math.sqrt(3) / 2) < 1e-12 and abs(J2[4, 4] - 0.5) < 1e-12


@pytest.mark.parametrize("t", TRIPLES)
def test_w3j_two_derivations_agree_and_are_invariant(t):
    C = so3.wigner_3j(*t)
    assert np.abs(C - oso3.w3j(*t)).max() < 1e-10
    assert abs(np.linalg.norm(C) - 1) < 1e-12
    a, b, c = 0.7, 1.3, 2.1
    D = [so3.wigner_D(l, a, b, c)[0] for l in t]
    assert np.abs(np.einsum('ijk,ai,bj,ck->abc', C, *D) - C).max() < 1e-12


def test_w3j_known_answers():
    for l in range(4):
        d = np.eye(2 * l + 1) / math.sqrt(2 * l + 1)
        assert np.abs(so3.wigner_3j(l, 0, l)[:, 0, :] - d).max() < 1e-12
        assert np.abs(so3.wigner_3j(0, l, l)[0] - d).max() < 1e-12
        assert np.abs(so3.wigner_3j(l, l, 0)[:, :, 0] - d).max() < 1e-12
    assert abs(so3.wigner_3j(1, 1, 1)[0, 1, 2] - 1 / math.sqrt(6)) < 1e-12
    nnz = {(0, 1, 1): 3, (0, 2, 2): 5, (1, 0, 1): 3, (1, 1, 0): 3, (1, 1, 1): 6, (1, 1, 2): 11, (1, 2, 1): 11, (1, 2, 2): 16,
           (2, 0, 2): 5, (2, 1, 1): 11, (2, 1, 2): 16, (2, 2, 0): 5, (2, 2, 1): 16, (2, 2, 2): 25}
    for t, n in nnz.items():
        assert int((so3.wigner_3j(*t) != 0).sum()) == n
    # sparse MAC count of the lmax-2 depth-wise TP quoted in SURVEY §8(c): 3424
    from diffusion_edf_amd.params import dtp_paths
    paths = dtp_paths([(64, 0), (32, 1), (16, 2)], [0, 1, 2], [1, 1, 1], [0, 1, 2])
    assert sum(m1 * int((so3.wigner_3j(l1, l2, l3) != 0).sum()) for l1, l2, l3, m1, _ in paths) == 3424
    assert sum(p[3] for p in paths) == 480


def test_normalize2mom_constants():
    """e3nn.math.normalize2mom: cst = E[f(z)^2]^-1/2 over 1e6 float64 normals from torch.Generator().manual_seed(0)."""
    gen = torch.Generator(device="cpu").manual_seed(0)
    z = torch.randn(1_000_000, generator=gen, dtype=torch.float64)

    def cst(f):
        return f(z).pow(2).mean().pow(-0.5).item()

    def slrelu(x, a=0.2):
        return ((1 + a) / 2) * x + ((1 - a) / 2) * x * (2 * torch.sigmoid(x) - 1)
    assert abs(cst(torch.nn.functional.silu) - so3.NORM2MOM_SILU) < 1e-12
    assert abs(cst(torch.sigmoid) - so3.NORM2MOM_SIGMOID) < 1e-12
    assert abs(cst(slrelu) - so3.NORM2MOM_SLRELU02) < 1e-12
    assert (oso3.C_SILU, oso3.C_SIGMOID, oso3.C_SLRELU) == (so3.NORM2MOM_SILU, so3.NORM2MOM_SIGMOID, so3.NORM2MOM_SLRELU02)


def test_generated_header_is_up_to_date():
    assert open(gen_tables.header_path()).read() == gen_tables.gen_header()
