"""Emit ``csrc/dedf_tables.h``: Clebsch–Gordan contractions as straight-line device code.

For every (l1, l2, l3) the kernels need, two functions are generated with the Wigner-3j coefficients (times the
e3nn path factor sqrt(2 l3 + 1), reference equiformer/tensor_product_rescale.py:38-42 with
path_normalization='none') folded into float literals:

    CG<l1,l2,l3>::make(y, m)      m[e]   = sum_j c_ijk y[j]        for every structurally non-zero (i,k) pair e
    CG<l1,l2,l3>::apply(x, m, o)  o[k]   = sum_i x[i] m[e(i,k)]
    CG<l1,l2,l3>::acc<I>(g, m, o) o[k]  += g m[e(I,k)]              for the k paired with component I   (output-side form:
                                                                    the contraction applied AFTER a linear map of component I)

so that  o[k] = sqrt(2l3+1) * sum_ij w3j[i,j,k] x[i] y[j].  Only non-zero terms are emitted (3 424 instead of
14 880 MACs per edge for the lmax-2 depth-wise TP).  Also emitted: J-matrix literals for the Wigner-D kernel and
the normalize2mom constants.

Run:  python -m diffusion_edf_amd.gen_tables   (the header is committed; tests check it is up to date)
"""
from __future__ import annotations

import math
import os

import numpy as np

from . import so2, so3

LMAX = 3


def _lit(v: float) -> str:
    t = f"{np.float32(v):.9g}"
    if "." not in t and "e" not in t:
        t += ".0"
    return t + "f"


def gen_cg(l1: int, l2: int, l3: int) -> str:
    C = so3.wigner_3j(l1, l2, l3) * math.sqrt(2 * l3 + 1)
    d1, d2, d3 = C.shape
    pairs = [(i, k) for i in range(d1) for k in range(d3) if np.any(C[i, :, k] != 0)]
    idx = {p: e for e, p in enumerate(pairs)}
    out = [f"template <> struct CG<{l1}, {l2}, {l3}> {{",
           f"    static constexpr int NM = {len(pairs)};   // non-zero (i,k) pairs; dense would be {d1 * d3}",
           f"    static constexpr int NNZ = {int((C != 0).sum())};",
           "    template <class Y, class M> DEDF_DEV static void make(const Y& y, M& m) {"]
    for (i, k), e in idx.items():
        terms = [f"{_lit(C[i, j, k])} * y[{j}]" for j in range(d2) if C[i, j, k] != 0]
        out.append(f"        m[{e}] = " + " + ".join(terms) + ";")
    out.append("    }")
    out.append("    template <class X, class M, class O> DEDF_DEV static void apply(const X& x, const M& m, O& o) {")
    for k in range(d3):
        terms = [f"x[{i}] * m[{idx[(i, k)]}]" for i in range(d1) if (i, k) in idx]
        out.append(f"        o[{k}] = " + (" + ".join(terms) if terms else "0.0f") + ";")
    out.append("    }")
    out.append("    template <int I, class M, class O> DEDF_DEV static void acc(const float g, const M& m, O& o) {")
    for i in range(d1):
        ks = [k for k in range(d3) if (i, k) in idx]
        body = " ".join(f"o[{k}] += g * m[{idx[(i, k)]}];" for k in ks)
        out.append(f"        if constexpr (I == {i}) {{ {body} }}")
    out.append("    }")
    out.append("};")
    return "\n".join(out)


def gen_so2_tables() -> str:
    """Edge-aligned-frame form of the depth-wise TPs (diffusion_edf_amd/so2.py): per path the (output component k, source component i,
    coefficient c) terms as constexpr tables shared by the host packers (which fold kSo2Ref into the linear layers' weights) and the
    kernels (which keep the compile-time ratio c / ref, +-1 for most terms)."""
    n = LMAX + 1
    NT = np.zeros((n, n, n), dtype=int)
    K = np.zeros((n, n, n, 7), dtype=int)
    I = np.zeros((n, n, n, 7), dtype=int)
    C = np.zeros((n, n, n, 7))
    R = np.ones((n, n, n))
    for l1 in range(n):
        for l2 in range(n):
            for l3 in range(abs(l1 - l2), min(LMAX, l1 + l2) + 1):
                t = so2.so2_terms(l1, l2, l3)
                NT[l1, l2, l3] = len(t)
                R[l1, l2, l3] = so2.so2_ref(l1, l2, l3)
                for e, (k, i, c) in enumerate(t):
                    K[l1, l2, l3, e], I[l1, l2, l3, e], C[l1, l2, l3, e] = k, i, c

    def arr(a, fmt):
        if a.ndim == 1:
            return "{" + ", ".join(fmt(v) for v in a) + "}"
        return "{" + ", ".join(arr(b, fmt) for b in a) + "}"
    out = ["// ---- edge-aligned-frame (SO(2)) form of the depth-wise TPs: out'[k] = c x'[i] per term, see diffusion_edf_amd/so2.py ----",
           f"constexpr int kSo2NT[{n}][{n}][{n}] = {arr(NT, lambda v: str(int(v)))};",
           f"constexpr int kSo2K[{n}][{n}][{n}][7] = {arr(K, lambda v: str(int(v)))};",
           f"constexpr int kSo2I[{n}][{n}][{n}][7] = {arr(I, lambda v: str(int(v)))};",
           f"constexpr float kSo2C[{n}][{n}][{n}][7] = {arr(C, _lit)};",
           f"constexpr float kSo2Ref[{n}][{n}][{n}] = {arr(R, _lit)};"]
    return "\n".join(out)


def gen_rot(l: int) -> str:
    """Straight-line device code of D^l(g) (rotate into the edge frame) and its transpose: X(gamma), J, X(beta), J as in so2.rot_in_program.
    T carries cos / sin of m gamma and m beta (m = 1 .. l) as cg[m-1], sg[m-1], cb[m-1], sb[m-1]."""
    d = 2 * l + 1
    J = so3.J_matrix(l)

    def stage(st):
        if st[0] == 'J':
            rows = []
            for r in range(d):
                terms = []
                for c in range(d):
                    if J[r, c] == 0:
                        continue
                    if abs(abs(J[r, c]) - 1.0) < 1e-14:
                        terms.append(("-" if J[r, c] < 0 else "") + f"v[{c}]")
                    else:
                        terms.append(f"{_lit(J[r, c])} * v[{c}]")
                rows.append(" + ".join(terms).replace("+ -", "- "))
            return "        { const float u[%d] = {%s}; %s }" % (d, ", ".join(rows), " ".join(f"v[{r}] = u[{r}];" for r in range(d)))
        which, sgn = st[1], st[2]
        parts = []
        for m in range(1, l + 1):
            i, j = l - m, l + m
            c, s = f"t.c{which}[{m - 1}]", f"t.s{which}[{m - 1}]"
            op1, op2 = ("+", "-") if sgn > 0 else ("-", "+")
            parts.append(f"{{ const float a = v[{i}], b = v[{j}]; v[{i}] = {c} * a {op1} {s} * b; v[{j}] = {c} * b {op2} {s} * a; }}")
        return "        " + " ".join(parts)
    out = [f"template <> struct Rot<{l}> {{",
           f"    template <class T> DEDF_DEV static void in(float (&v)[{d}], const T& t) {{"]
    out += [stage(st) for st in so2.rot_in_program(l)]
    out += ["    }", f"    template <class T> DEDF_DEV static void out(float (&v)[{d}], const T& t) {{"]
    out += [stage(st) for st in so2.rot_out_program(l)]
    out += ["    }", "};"]
    return "\n".join(out)


def gen_header() -> str:
    parts = ["// GENERATED by diffusion_edf_amd/gen_tables.py — do not edit.",
             "// Wigner-3j (real basis, e3nn construction) x sqrt(2 l3 + 1), J matrices, normalize2mom constants.",
             "#pragma once",
             "#ifndef DEDF_DEV",
             "#define DEDF_DEV __host__ __device__ __forceinline__",
             "#endif",
             "namespace dedf {",
             f"constexpr float kNormSilu = {_lit(so3.NORM2MOM_SILU)};",
             f"constexpr float kNormSigmoid = {_lit(so3.NORM2MOM_SIGMOID)};",
             f"constexpr float kNormSlrelu = {_lit(so3.NORM2MOM_SLRELU02)};",
             "template <int L1, int L2, int L3> struct CG;"]
    for l1 in range(LMAX + 1):
        for l2 in range(LMAX + 1):
            for l3 in range(abs(l1 - l2), min(LMAX, l1 + l2) + 1):
                parts.append(gen_cg(l1, l2, l3))
    for l in range(1, LMAX + 1):
        J = so3.J_matrix(l)
        d = 2 * l + 1
        rows = ", ".join("{" + ", ".join(_lit(J[i, j]) for j in range(d)) + "}" for i in range(d))
        parts.append(f"constexpr float kJ{l}[{d}][{d}] = {{{rows}}};")
    parts.append(gen_so2_tables())
    parts.append("template <int L> struct Rot;")
    parts.append("template <> struct Rot<0> { template <class T> DEDF_DEV static void in(float (&)[1], const T&) {} template <class T> DEDF_DEV static void out(float (&)[1], const T&) {} };")
    for l in range(1, LMAX + 1):
        parts.append(gen_rot(l))
    parts.append("}  // namespace dedf")
    return "\n".join(parts) + "\n"


def header_path() -> str:
    return os.path.join(os.path.dirname(__file__), "csrc", "dedf_tables.h")


def main():
    with open(header_path(), "w") as f:
        f.write(gen_header())
    print("wrote", header_path())


if __name__ == "__main__":
    main()
