"""Edge-aligned-frame ("SO(2)") form of the depth-wise tensor products (diffusion_edf_amd/so2.py, dedf_tables.h::kSo2* / Rot<l>,
dedf_net.h::make_dtp_walk_so2 / make_sval_walk): the float64 model against the reference form of the TP
(equiformer/tensor_product_rescale.py:352-382 restated in so3.py), the generated device code compiled for the host, and the
kernel's walks against the term tables."""
import json
import math
import os
import shutil
import subprocess

import numpy as np
import pytest

from diffusion_edf_amd import so2, so3

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "diffusion_edf_amd", "csrc")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def _dirs(n=40, seed=0):
    rng = np.random.default_rng(seed)
    r = rng.normal(size=(n, 3))
    r /= np.linalg.norm(r, axis=-1, keepdims=True)
    r[0] = [0, 1, 0]          # the polar axis itself and its opposite: rho = 0, gamma is free
    r[1] = [0, -1, 0]
    r[2] = [1, 0, 0]
    r[3] = [0, 0, -1]
    return rng, r


def test_frame_takes_the_edge_to_the_polar_axis():
    _, r = _dirs()
    for l in (1, 2, 3):
        D = so2.rot_in_matrix(l, r)
        Y = so3.spherical_harmonics(l, r)
        tgt = np.zeros(2 * l + 1)
        tgt[l] = math.sqrt(2 * l + 1)
        assert np.abs(np.einsum('nij,nj->ni', D, Y) - tgt).max() < 1e-13
        assert np.abs(np.einsum('nij,nkj->nik', D, D) - np.eye(2 * l + 1)).max() < 1e-13


def test_rotation_programs_are_the_wigner_matrices():
    rng, r = _dirs()
    for l in (1, 2, 3):
        x = rng.normal(size=(len(r), 2 * l + 1))
        D = so2.rot_in_matrix(l, r)
        xin = so2.run_program(l, so2.rot_in_program(l), x, r)
        assert np.abs(xin - np.einsum('nij,nj->ni', D, x)).max() < 1e-13
        assert np.abs(so2.run_program(l, so2.rot_out_program(l), xin, r) - x).max() < 1e-13


@pytest.mark.parametrize("L", [1, 2, 3])
def test_every_path_in_the_edge_frame_equals_the_reference_form(L):
    rng, r = _dirs(seed=L)
    for l1 in range(L + 1):
        for l2 in range(L + 1):
            for l3 in range(abs(l1 - l2), min(L, l1 + l2) + 1):
                x = rng.normal(size=(len(r), 2 * l1 + 1))
                for cns in (1.0, 0.37, 0.0):
                    ref = so2.dtp_direct(l1, l2, l3, x, r, cns)
                    xp = so2.run_program(l1, so2.rot_in_program(l1), x, r)
                    out = so2.run_program(l3, so2.rot_out_program(l3), so2.dtp_edge_frame(l1, l2, l3, xp, cns), r)
                    assert np.abs(out - ref).max() < 1e-12, (l1, l2, l3, cns)
                # one source component per output component, |m| preserved, +-m coefficients equal (even paths) / opposite (odd paths)
                terms = {k - l3: (i - l1, c) for k, i, c in so2.so2_terms(l1, l2, l3)}
                for m, (mi, c) in terms.items():
                    if (l1 + l2 + l3) % 2 == 0:
                        assert mi == m and abs(terms[-m][1] - c) < 1e-14
                    else:
                        assert mi == -m and m != 0 and abs(terms[-m][1] + c) < 1e-14


def test_two_chained_tensor_products_stay_in_the_edge_frame():
    """graph_attention.py:231-247 per edge: DTP -> per-degree linear -> gate by scalars -> DTP: one rotate-in, one rotate-out."""
    rng, r = _dirs(n=12, seed=5)
    L, mul = 2, 3
    paths = [(l1, l2, l3) for l1 in range(L + 1) for l2 in range(L + 1) for l3 in range(abs(l1 - l2), L + 1) if l3 <= l1 + l2]
    for n in range(len(r)):
        x = [rng.normal(size=(mul, 2 * l + 1)) for l in range(L + 1)]
        w1 = {p: rng.normal(size=mul) for p in paths}
        w2 = {p: rng.normal(size=mul) for p in paths}
        lin = {p: rng.normal(size=(mul, mul)) for p in paths}
        gate = rng.normal(size=(L + 1, mul))

        def chain(feat, edge_frame):
            def dtp(f, w):
                out = [np.zeros((mul, 2 * l + 1)) for l in range(L + 1)]
                for (l1, l2, l3) in paths:
                    t = so2.dtp_edge_frame(l1, l2, l3, f[l1], 0.6) if edge_frame else so2.dtp_direct(l1, l2, l3, f[l1], r[n], 0.6)
                    out[l3] += lin[(l1, l2, l3)].T @ (w[(l1, l2, l3)][:, None] * t)
                return out
            a = dtp(feat, w1)
            a = [a[l] * gate[l][:, None] for l in range(L + 1)]
            return dtp(a, w2)
        ref = chain(x, False)
        xp = [so2.run_program(l, so2.rot_in_program(l), x[l], r[n]) for l in range(L + 1)]
        got = chain(xp, True)
        got = [so2.run_program(l, so2.rot_out_program(l), got[l], r[n]) for l in range(L + 1)]
        for l in range(L + 1):
            assert np.abs(got[l] - ref[l]).max() < 1e-11


_ROT_SRC = r"""
#include <cstdio>
#include <cmath>
#define DEDF_DEV inline
#include "dedf_tables.h"
using namespace dedf;
struct T3 { float cg[3], sg[3], cb[3], sb[3]; };
int main() {
    double r[3]; float v1[3], v2[5], v3[7];
    while (scanf("%lf %lf %lf", &r[0], &r[1], &r[2]) == 3) {
        for (int i = 0; i < 3; ++i) if (scanf("%f", &v1[i]) != 1) return 1;
        for (int i = 0; i < 5; ++i) if (scanf("%f", &v2[i]) != 1) return 1;
        for (int i = 0; i < 7; ++i) if (scanf("%f", &v3[i]) != 1) return 1;
        const float x = r[0], y = r[1], z = r[2], rho = sqrtf(x * x + z * z), inv = rho > 0 ? 1.0f / rho : 0.0f;
        T3 t;
        t.cg[0] = rho > 0 ? z * inv : 1.0f; t.sg[0] = rho > 0 ? -x * inv : 0.0f; t.cb[0] = y; t.sb[0] = -rho;
        for (int m = 1; m < 3; ++m) {
            t.cg[m] = t.cg[m - 1] * t.cg[0] - t.sg[m - 1] * t.sg[0]; t.sg[m] = t.sg[m - 1] * t.cg[0] + t.cg[m - 1] * t.sg[0];
            t.cb[m] = t.cb[m - 1] * t.cb[0] - t.sb[m - 1] * t.sb[0]; t.sb[m] = t.sb[m - 1] * t.cb[0] + t.cb[m - 1] * t.sb[0];
        }
        Rot<1>::in(v1, t); Rot<2>::in(v2, t); Rot<3>::in(v3, t);
        for (float f : v1) printf("%.9g ", f); for (float f : v2) printf("%.9g ", f); for (float f : v3) printf("%.9g ", f);
        Rot<1>::out(v1, t); Rot<2>::out(v2, t); Rot<3>::out(v3, t);
        for (float f : v1) printf("%.9g ", f); for (float f : v2) printf("%.9g ", f); for (float f : v3) printf("%.9g ", f);
        printf("\n");
    }
    return 0;
}
"""

_WALK_SRC = r"""
#include <cstdio>
#define DEDF_DEV inline
#include "dedf_net.h"
using namespace dedf;
template <int L> void show() {
    printf("{\"L\": %d, \"walk\": [", L);
    for (int p = 0; p < dtp_wn<L>() / 16; ++p) { auto pi = dtp_pos_path<L, true>(p);
        printf("%s[%d, %d, %d, %d, %d]", p ? ", " : "", dtp_pos_chunk<L, true>(p), pi.l1, pi.l2, pi.l3, (int)dtp_pos_same_x<L, true>(p, p - 1)); }
    printf("], \"slots\": [%d, %d], \"items\": [", dtp_num_slots<L, true>(r0_tiles<L>()), dtp_num_slots<L, false>(r0_tiles<L>()));
    for (int I = 0; I < sval_num_items<L>(); ++I) { auto it = sval_item<L>(I); auto pi = dtp_path<L>(it.p);
        printf("%s{\"path\": [%d, %d, %d], \"c\": %d, \"set\": %d, \"coef\": %.9g, \"ge\": %d, \"ops\": [", I ? ", " : "", pi.l1, pi.l2, pi.l3, it.c, it.set, it.coef, it.group_end);
        for (int a = 0; a < it.na; ++a) printf("%s[%d, %d, %d, %d, %d, %d, %d]", a ? ", " : "", it.acc[a], it.bq[a], (int)it.neg[a], it.aslot[a], (int)it.first[a], it.tile[a], (int)it.up[a]);
        printf("]}"); }
    printf("], \"n_slots\": %d, \"park\": [", sval_num_slots<L>());
    for (int l = 0; l <= L; ++l) printf("%s%d", l ? ", " : "", park_slot<L>(l, 0, 0));
    printf("]}\n");
}
int main() { show<1>(); show<2>(); show<3>(); return 0; }
"""


def _host_build(tmp_path, name, src, compiler):
    cpp = tmp_path / (name + ".cpp")
    cpp.write_text(src)
    exe = tmp_path / name
    subprocess.run([compiler, "-std=c++20", "-O1", "-w", "-I", CSRC, str(cpp), "-o", str(exe)], check=True)
    return str(exe)


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs a host compiler")
def test_generated_rotation_code_on_the_host(tmp_path):
    exe = _host_build(tmp_path, "rot", _ROT_SRC, "g++")
    rng, r = _dirs(n=30, seed=7)
    v = rng.normal(size=(len(r), 15))
    inp = "\n".join(" ".join(f"{a:.9g}" for a in list(r[n]) + list(v[n])) for n in range(len(r)))
    out = subprocess.run([exe], input=inp, capture_output=True, text=True, check=True).stdout
    o = np.array([[float(t) for t in ln.split()] for ln in out.strip().splitlines()])
    ref = np.concatenate([so2.run_program(l, so2.rot_in_program(l), v[:, s:s + 2 * l + 1], r) for l, s in ((1, 0), (2, 3), (3, 8))], axis=1)
    assert np.abs(o[:, :15] - ref).max() < 5e-6          # fp32 device arithmetic against the float64 model
    assert np.abs(o[:, 15:] - v).max() < 5e-6            # out(in(v)) = v


@pytest.mark.skipif(not os.path.exists(CLANG), reason="needs the ROCm clang (for _Float16 in dedf_layout.h)")
def test_kernel_walks_cover_every_term_once(tmp_path):
    exe = _host_build(tmp_path, "walk", _WALK_SRC, CLANG)
    out = subprocess.run([exe], capture_output=True, text=True, check=True).stdout
    for ln in out.strip().splitlines():
        W = json.loads(ln)
        L = W["L"]
        mul = lambda l: 16 if l >= 3 else 64 >> l
        paths = [(l1, l2, l3) for l1 in range(L + 1) for l2 in range(L + 1) for l3 in range(abs(l1 - l2), min(L, l1 + l2) + 1)]
        n_chunks = sum(mul(p[0]) // 16 for p in paths)
        # stage 1: every 16-channel chunk once, scalar outputs first, then the l3 >= 1 chunks by (l1, channel range)
        chunks = [w[0] for w in W["walk"]]
        assert sorted(chunks) == list(range(n_chunks))
        l3s = [w[3] for w in W["walk"]]
        n0 = sum(1 for v in l3s if v == 0)
        assert all(v == 0 for v in l3s[:n0]) and all(v >= 1 for v in l3s[n0:])
        # ... in groups of output degrees (lmax <= 2: one group; lmax 3: {1, 2} then {3}), each walked by (l1, channel range)
        grp = lambda l3: 0 if l3 == 0 else (2 if (L == 3 and l3 == 3) else 1)
        gs = [grp(w[3]) for w in W["walk"]]
        assert gs == sorted(gs)
        for g in set(gs):
            l1s = [w[1] for w in W["walk"] if grp(w[3]) == g]
            assert l1s == sorted(l1s)
        assert W["slots"][0] == W["slots"][1]          # one A slot per chunk, as in the general form
        # value: the items' terms are exactly the edge-frame terms of every (path, K-chunk), coefficient = folded class constant x sign
        park0 = W["park"]
        got = {}
        seen_first = set()
        for it in W["items"]:
            l1, l2, l3 = it["path"]
            assert it["set"] == (1 if l2 > 0 else 0)
            assert len(it["ops"]) <= (4 if L == 3 else 5)
            for acc, bq, neg, slot, first, tile, up in it["ops"]:
                paired = L == 3 and l3 >= 2
                assert (tile, up) == ((acc // 2, acc % 2) if paired else (acc, 0))
                if l3 == 0:
                    k, i = 0, l1
                    comp = (bq - park0[l1]) // (mul(l1) // 16)
                    assert comp == i
                    got.setdefault((l1, l2, l3, it["c"], 0), []).append(it["coef"])
                else:
                    k = acc
                    comp = (bq - park0[l1]) // (mul(l1) // 16)
                    got.setdefault((l1, l2, l3, it["c"], k), []).append((comp, -it["coef"] if neg else it["coef"]))
                    assert (bq - park0[l1]) % (mul(l1) // 16) == it["c"]
                key = (l3, it["set"], tile)
                assert bool(first) == (key not in seen_first)
                seen_first.add(key)
        for (l1, l2, l3) in paths:
            terms = so2.so2_terms(l1, l2, l3)
            for c in range(mul(l1) // 16):
                if l3 == 0:
                    assert len(got[(l1, l2, l3, c, 0)]) == 2 and abs(got[(l1, l2, l3, c, 0)][0] - terms[0][2]) < 1e-6
                    continue
                for k, i, cf in terms:
                    e = got.pop((l1, l2, l3, c, k))
                    assert len(e) == 1 and e[0][0] == i and abs(e[0][1] - cf) < 1e-6, (l1, l2, l3, c, k, e, cf)
        assert not [k for k in got if k[2] != 0]
        ends = [it["ge"] for it in W["items"] if it["ge"] >= 0]
        assert ends == list(range(L, 0, -1)) + [0]          # highest degree first, the scalars last (dedf_net.h::make_sval_walk)
