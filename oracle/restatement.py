"""ORACLE — CPU restatement of the Diffusion-EDF score-head hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this file.
The product (``diffusion_edf_amd``) never does; it fails loudly when its HIP library is missing.

It mirrors the reference op for op, materialising the same per-edge tensors (that is also why it is the
"reference CPU path" that bench.py times).  Citations are relative to /root/reference:

  score_head.py:142-211 -> gnn_data.py:80-113 -> wigner.py:17-81,119-134,203-283 (+ transforms.py:83-110,
  198-208,230-308) -> multiscale_tensor_field.py:192-260 -> graph_parser.py:146-224,272-286,336-345
  (+ radial_func.py:10-70,168-227,291-316, irreps_utils.py:20-63) -> gnn_block.py:164-218,51-57 ->
  graph_attention.py:218-273 -> equiformer/graph_attention_transformer.py:60-201 ->
  equiformer/tensor_product_rescale.py:20-185,188-268,352-392 -> equiformer/fast_activation.py:14-23,31-152,
  156-224 -> equiformer/layer_norm.py:64-156 -> equiformer/radial_func.py:11-60 -> score_model_base.py:110-204.

Third-party pieces that are NOT in /root/reference and are restated from their published algorithm:
  e3nn==0.4.4 (setup.py:28): o3.TensorProduct ('uvu'/'uvw', irrep_normalization='component',
  path_normalization='none' -> path coefficient sqrt(2 l_out + 1)), o3.SphericalHarmonics, o3.wigner_3j,
  normalize2mom, ElementwiseTensorProduct; torch_cluster.radius; torch_scatter.scatter / scatter_logsumexp;
  edf_interface.data.pcd_utils.transform_points.

PARITY STATUS: the reference ships no tests, golden tensors or weights for this path (all *.pt are LFS
stubs) and its dependencies cannot be imported here, so end-to-end parity is **unpinned**.  What is pinned:
the two importable reference modules (transforms.py, radial_func.py) via tests/golden/*.npz, and the
equivariance invariants of SURVEY §8(c).
"""
from __future__ import annotations

import math
from typing import Dict, List, NamedTuple, Optional, Sequence, Tuple

import numpy as np
import torch

from . import so3_oracle as so3

Tensor = torch.Tensor
Irreps = List[Tuple[int, int]]  # [(mul, l)], even parity only


# =================================================================================================
# irreps bookkeeping
# =================================================================================================

def parse_irreps(s) -> Irreps:
    if isinstance(s, (list, tuple)):
        return [(int(m), int(l)) for m, l in s]
    out = []
    for tok in str(s).replace(' ', '').split('+'):
        mul, ir = tok.split('x') if 'x' in tok else ('1', tok)
        assert ir[-1] == 'e', s
        out.append((int(mul), int(ir[:-1])))
    return out


def simplify(irreps: Irreps) -> Irreps:
    out: Irreps = []
    for mul, l in irreps:
        if out and out[-1][1] == l:
            out[-1] = (out[-1][0] + mul, l)
        elif mul > 0:
            out.append((mul, l))
    return out


def dim(irreps: Irreps) -> int:
    return sum(m * (2 * l + 1) for m, l in irreps)


def slices(irreps: Irreps) -> List[Tuple[int, int]]:
    out, s = [], 0
    for m, l in irreps:
        out.append((s, s + m * (2 * l + 1)))
        s += m * (2 * l + 1)
    return out


def sort_even_first(irreps: Irreps):
    """tensor_product_rescale.py:385-392 (all parities even here -> stable sort on l)."""
    order = sorted(range(len(irreps)), key=lambda i: (irreps[i][1], i))
    inv = tuple(order)
    p = [0] * len(irreps)
    for new, old in enumerate(inv):
        p[old] = new
    return [irreps[i] for i in order], p, inv


def irreps2gate(irreps: Irreps):
    """tensor_product_rescale.py:188-233."""
    scalars = simplify([(m, l) for m, l in irreps if l == 0])
    gated = simplify([(m, l) for m, l in irreps if l != 0])
    gates = simplify([(m, 0) for m, _ in gated])
    return scalars, gates, gated


# =================================================================================================
# e3nn restatements
# =================================================================================================

def w3j(l1, l2, l3, dtype) -> Tensor:
    return torch.tensor(so3.w3j(l1, l2, l3), dtype=dtype)


def spherical_harmonics(irreps_sh: Irreps, vec: Tensor) -> Tensor:
    """o3.SphericalHarmonics(normalize=True, normalization='component') — graph_parser.py:135."""
    n = vec.norm(dim=-1, keepdim=True)
    u = vec / torch.clamp(n, min=1e-12)          # F.normalize
    x, y, z = u[..., 0], u[..., 1], u[..., 2]
    out = []
    for mul, l in irreps_sh:
        assert mul == 1
        if l == 0:
            out.append(torch.ones_like(x)[..., None])
        elif l == 1:
            out.append(math.sqrt(3.0) * torch.stack([x, y, z], dim=-1))
        elif l == 2:
            s3 = math.sqrt(3.0)
            rho = x * x + z * z
            out.append(math.sqrt(5.0) * torch.stack(
                [s3 * x * z, s3 * x * y, y * y - rho / 2, s3 * y * z, s3 * (z * z - x * x) / 2], dim=-1))
        else:
            out.append(torch.tensor(so3.sh(l, u.double().numpy()), dtype=vec.dtype))
    return torch.cat(out, dim=-1)


class TPInstr(NamedTuple):
    i1: int
    i2: int
    iout: int
    mode: str          # 'uvu' | 'uvw'


class TensorProduct:
    """o3.TensorProduct(..., path_normalization='none') as wrapped by TensorProductRescale
    (tensor_product_rescale.py:20-152): out[slot] += sqrt(2 l_out+1) * sum_ij w3j_ijk (x1 (x) x2)."""

    def __init__(self, ir1: Irreps, ir2: Irreps, irout: Irreps, instr: Sequence[TPInstr]):
        self.ir1, self.ir2, self.irout, self.instr = ir1, ir2, irout, list(instr)
        self.s1, self.s2, self.so = slices(ir1), slices(ir2), slices(irout)
        self.shapes = []
        for ins in self.instr:
            m1, m2, mo = ir1[ins.i1][0], ir2[ins.i2][0], irout[ins.iout][0]
            self.shapes.append((m1, m2) if ins.mode == 'uvu' else (m1, m2, mo))
        self.weight_numel = sum(int(np.prod(s)) for s in self.shapes)

    def __call__(self, x1: Tensor, x2: Tensor, weight: Tensor) -> Tensor:
        """weight: (weight_numel,) shared or (Z, weight_numel) per row."""
        Z = x1.shape[0]
        out = x1.new_zeros(Z, dim(self.irout))
        off = 0
        for ins, shp in zip(self.instr, self.shapes):
            n = int(np.prod(shp))
            w = weight[..., off:off + n].reshape(weight.shape[:-1] + shp)
            off += n
            (m1, l1), (m2, l2), (mo, lo) = self.ir1[ins.i1], self.ir2[ins.i2], self.irout[ins.iout]
            a = x1[:, self.s1[ins.i1][0]:self.s1[ins.i1][1]].reshape(Z, m1, 2 * l1 + 1)
            b = x2[:, self.s2[ins.i2][0]:self.s2[ins.i2][1]].reshape(Z, m2, 2 * l2 + 1)
            C = w3j(l1, l2, lo, x1.dtype) * math.sqrt(2 * lo + 1)
            xx = torch.einsum('zui,zvj->zuvij', a, b)
            if ins.mode == 'uvu':
                if w.dim() == 2:
                    r = torch.einsum('uv,ijk,zuvij->zuk', w, C, xx)
                else:
                    r = torch.einsum('zuv,ijk,zuvij->zuk', w, C, xx)
            else:
                if w.dim() == 3:
                    r = torch.einsum('uvw,ijk,zuvij->zwk', w, C, xx)
                else:
                    r = torch.einsum('zuvw,ijk,zuvij->zwk', w, C, xx)
            s, e = self.so[ins.iout]
            out[:, s:e] += r.reshape(Z, e - s)          # (explicit width: Z may be 0 when no edge exists)
        return out


def depthwise_tp(ir_in: Irreps, ir_sh: Irreps, ir_node_out: Irreps) -> TensorProduct:
    """DepthwiseTensorProduct — tensor_product_rescale.py:352-382 (weight blocks stay in creation order,
    outputs re-indexed by sort_irreps_even_first)."""
    out_ls = {l for _, l in ir_node_out}
    irreps_output, instr = [], []
    for i, (mul, l1) in enumerate(ir_in):
        for j, (_, l2) in enumerate(ir_sh):
            for lo in range(abs(l1 - l2), l1 + l2 + 1):
                if lo in out_ls or lo == 0:
                    k = len(irreps_output)
                    irreps_output.append((mul, lo))
                    instr.append((i, j, k))
    irreps_sorted, p, _ = sort_even_first(irreps_output)
    instr = [TPInstr(i, j, p[k], 'uvu') for i, j, k in instr]
    return TensorProduct(ir_in, ir_sh, irreps_sorted, instr)


def fctp(ir1: Irreps, ir2: Irreps, irout: Irreps) -> TensorProduct:
    """FullyConnectedTensorProductRescale — tensor_product_rescale.py:155-173."""
    instr = [TPInstr(i1, i2, io, 'uvw')
             for i1, (_, l1) in enumerate(ir1) for i2, (_, l2) in enumerate(ir2)
             for io, (_, lo) in enumerate(irout) if abs(l1 - l2) <= lo <= l1 + l2]
    return TensorProduct(ir1, ir2, irout, instr)


def add_bias(out: Tensor, irout: Irreps, P: Dict[str, Tensor], prefix: str) -> Tensor:
    """TensorProductRescale.forward_tp_rescale_bias — tensor_product_rescale.py:137-147: one bias
    Parameter per 0e slice of irreps_out.simplify(), named `<prefix>.bias.<n>`."""
    simp = simplify(irout)
    n = 0
    for (mul, l), (s, e) in zip(simp, slices(simp)):
        if l == 0:
            key = f"{prefix}.bias.{n}"
            if key in P:
                out[:, s:e] += P[key]
            n += 1
    return out


def linear_rs(x: Tensor, ir_in: Irreps, ir_out: Irreps, P, prefix: str, bias: bool = True) -> Tensor:
    """LinearRS — tensor_product_rescale.py:176-185."""
    tp = fctp(ir_in, [(1, 0)], ir_out)
    y = torch.ones_like(x[:, 0:1])
    out = tp(x, y, P[f"{prefix}.tp.weight"])
    if bias:
        out = add_bias(out, ir_out, P, prefix)
    return out


def silu_n(x):      # normalize2mom(SiLU)
    return torch.nn.functional.silu(x) * so3.C_SILU


def sigmoid_n(x):   # normalize2mom(sigmoid)
    return torch.sigmoid(x) * so3.C_SIGMOID


def smooth_leaky_relu_n(x, alpha=0.2):
    """fast_activation.py:14-23 wrapped by normalize2mom (Activation, :69)."""
    x1 = ((1 + alpha) / 2) * x
    x2 = ((1 - alpha) / 2) * x * (2 * torch.sigmoid(x) - 1)
    return (x1 + x2) * so3.C_SLRELU


def gate(x: Tensor, scalars: Irreps, gates: Irreps, gated: Irreps) -> Tensor:
    """fast_activation.py:156-224 with SiLU on scalars, sigmoid on gates."""
    ns, ng = dim(scalars), dim(gates)
    s = silu_n(x[:, :ns])
    if ng == 0:
        return s
    g = sigmoid_n(x[:, ns:ns + ng])
    v = x[:, ns + ng:]
    outs, gi, vi = [], 0, 0
    for mul, l in gated:
        d = 2 * l + 1
        blk = v[:, vi:vi + mul * d].reshape(-1, mul, d) * g[:, gi:gi + mul, None]
        outs.append(blk.reshape(-1, mul * d))
        gi += mul
        vi += mul * d
    return torch.cat([s] + outs, dim=-1)


def equivariant_layer_norm_v2(x: Tensor, irreps: Irreps, P, prefix: str, eps: float = 1e-5) -> Tensor:
    """EquivariantLayerNormV2.forward — equiformer/layer_norm.py:91-156 ('component', affine)."""
    fields, ix, iw, ib = [], 0, 0, 0
    for mul, l in irreps:
        d = 2 * l + 1
        f = x[:, ix:ix + mul * d].reshape(-1, mul, d)
        ix += mul * d
        if l == 0:
            f = f - f.mean(dim=1, keepdim=True)
        norm = f.pow(2).mean(-1).mean(dim=1, keepdim=True)
        norm = (norm + eps).pow(-0.5)
        norm = norm * P[f"{prefix}.affine_weight"][None, iw:iw + mul]
        iw += mul
        f = f * norm.reshape(-1, mul, 1)
        if d == 1:
            f = f + P[f"{prefix}.affine_bias"][ib:ib + mul].reshape(mul, 1)
            ib += mul
        fields.append(f.reshape(-1, mul * d))
    return torch.cat(fields, dim=-1)


def radial_profile(x: Tensor, P, prefix: str, n_layers: int) -> Tensor:
    """RadialProfile — equiformer/radial_func.py:11-60: (Linear, LayerNorm, SiLU) x (n-1), Linear(no bias)
    + offset.  Module indices in `net`: 0,1,2 / 3,4,5 / 6."""
    idx = 0
    for i in range(n_layers):
        W = P[f"{prefix}.net.{idx}.weight"]
        last = (i == n_layers - 1)
        x = x @ W.t()
        if not last:
            x = x + P[f"{prefix}.net.{idx}.bias"]
            x = torch.nn.functional.layer_norm(x, (x.shape[-1],), P[f"{prefix}.net.{idx + 1}.weight"],
                                               P[f"{prefix}.net.{idx + 1}.bias"], 1e-5)
            x = torch.nn.functional.silu(x)
            idx += 3
    return x + P[f"{prefix}.offset"].reshape(1, -1)


def vec2heads(x: Tensor, irreps_head: Irreps, H: int) -> Tensor:
    """Vec2AttnHeads — graph_attention_transformer.py:139-168."""
    N = x.shape[0]
    out, s = [], 0
    for mul, l in irreps_head:
        w = mul * H * (2 * l + 1)
        out.append(x[:, s:s + w].reshape(N, H, w // H))
        s += w
    return torch.cat(out, dim=2)


def heads2vec(x: Tensor, irreps_head: Irreps) -> Tensor:
    """AttnHeads2Vec — graph_attention_transformer.py:177-201."""
    N = x.shape[0]
    out, s = [], 0
    for mul, l in irreps_head:
        w = mul * (2 * l + 1)
        out.append(x[:, :, s:s + w].reshape(N, -1))
        s += w
    return torch.cat(out, dim=1)


# =================================================================================================
# geometry: transforms.py / wigner.py / radial_func.py restatements
# =================================================================================================

def quaternion_raw_multiply(a, b):      # transforms.py:113-129
    aw, ax, ay, az = torch.unbind(a, -1)
    bw, bx, by, bz = torch.unbind(b, -1)
    return torch.stack((aw * bw - ax * bx - ay * by - az * bz,
                        aw * bx + ax * bw + ay * bz - az * by,
                        aw * by - ax * bz + ay * bw + az * bx,
                        aw * bz + ax * by - ay * bx + az * bw), -1)


def quaternion_invert(q):               # transforms.py:132-144
    return q * torch.tensor([1, -1, -1, -1], dtype=q.dtype)


def quaternion_apply(q, p):             # transforms.py:147-165
    pq = torch.cat((p.new_zeros(p.shape[:-1] + (1,)), p), -1)
    return quaternion_raw_multiply(quaternion_raw_multiply(q, pq), quaternion_invert(q))[..., 1:]


def quaternion_to_matrix(q):            # transforms.py:83-110
    r, i, j, k = torch.unbind(q, -1)
    two_s = 2.0 / (q * q).sum(-1)
    o = torch.stack((1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                     two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                     two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1)
    return o.reshape(q.shape[:-1] + (3, 3))


def standardize_quaternion(q):          # transforms.py:198-208
    return torch.where(q[..., 0:1] < 0, -q, q)


def matrix_to_euler_yxy(M):
    """transforms.matrix_to_euler_angles(M, "YXY") — transforms.py:271-308 unrolled for this convention
    (keeps the signed-zero behaviour: identity -> (0, 0, pi))."""
    central = torch.acos(M[..., 1, 1])
    col = M[..., 1]                      # matrix[..., i2]  (i2 = 1)  -> M[..., :, 1]
    a0 = torch.atan2(col[..., 0], col[..., 2])
    row = M[..., 1, :]
    a2 = torch.atan2(row[..., 0], -row[..., 2])
    return torch.stack((a0, central, a2), -1)


def _z_rot_mat(angle: Tensor, l: int) -> Tensor:        # wigner.py:21-42
    M = angle.new_zeros((len(angle), 2 * l + 1, 2 * l + 1))
    inds = torch.arange(0, 2 * l + 1)
    rev = torch.arange(2 * l, -1, -1)
    freq = torch.arange(l, -l - 1, -1, dtype=angle.dtype)
    M[:, inds, rev] = torch.sin(freq * angle[:, None])
    M[:, inds, inds] = torch.cos(freq * angle[:, None])
    return M


def wigner_D(l, alpha, beta, gamma) -> Tensor:          # wigner.py:44-81
    Jm = torch.tensor(so3.J(l), dtype=alpha.dtype)
    return _z_rot_mat(alpha, l) @ Jm @ _z_rot_mat(beta, l) @ Jm @ _z_rot_mat(gamma, l)


def transform_feature_quaternion(irreps: Irreps, feature: Tensor, q: Tensor) -> Tensor:
    """TransformFeatureQuaternion.forward — wigner.py:257-283."""
    q = standardize_quaternion(q / torch.norm(q, dim=-1, keepdim=True))
    ang = matrix_to_euler_yxy(quaternion_to_matrix(q)).T
    alpha, beta, gamma = ang[0], ang[1], ang[2]
    outs = []
    for (mul, l), (s, e) in zip(irreps, slices(irreps)):
        f = feature[:, s:e]
        if l == 0:
            outs.append(f.expand(len(alpha), len(f), e - s))
        else:
            D = wigner_D(l, alpha, beta, gamma)
            f3 = f.reshape(f.shape[0], -1, 2 * l + 1)
            t = torch.einsum('tij,qmj->tqmi', D, f3)
            outs.append(t.reshape(t.shape[0], t.shape[1], -1))
    return torch.cat(outs, dim=-1)


def transform_points(points: Tensor, Ts: Tensor) -> Tensor:
    """edf_interface.data.pcd_utils.transform_points (un-vendored submodule; restated from its use at
    gnn_data.py:95): R(q) x + t for every pose."""
    return quaternion_apply(Ts[..., None, :4], points) + Ts[..., None, 4:]


def soft_step(x, n: int = 3):           # radial_func.py:15-17
    return (x > 0) * ((x < 1) * ((n + 1) * x.pow(n) - n * x.pow(n + 1)) + (x >= 1))


def soft_square_cutoff_2(x, ranges, n: int = 3):    # radial_func.py:31-70
    if ranges is None:
        return x
    left_end, left_begin, right_begin, right_end = ranges
    div_l = 1. if (left_end is None or left_begin is None) else left_begin - left_end
    div_r = 1. if (right_end is None or right_begin is None) else right_end - right_begin
    if right_begin is not None and left_end is None:
        return 1 - soft_step((x - right_begin) / div_r, n=n)
    if left_end is not None and right_begin is None:
        return soft_step((x - left_end) / div_l, n=n)
    if right_begin is not None and left_end is not None and left_begin is not None:
        mid = 0.5 * (left_begin + right_begin)
        return (1 - soft_step((x - right_begin) / div_r, n=n)) * (x > mid) + \
            soft_step((x - left_end) / div_l, n=n) * (x <= mid)
    return torch.ones_like(x)


def sinusoidal_embedding(x: Tensor, dim_: int, max_val: float, n: float) -> Tensor:   # radial_func.py:291-316
    x = x / max_val * n
    half = dim_ // 2
    emb = math.log(n) / (half - 1)
    emb = torch.exp(torch.arange(half, dtype=x.dtype) * -emb)
    emb = x[..., None] * emb
    return torch.cat((emb.sin(), emb.cos()), dim=-1)


def gaussian_radial_basis(dist: Tensor, P, prefix: str, dim_: int, max_val: float) -> Tensor:
    """GaussianRadialBasis.forward + _GaussianParamModule — radial_func.py:168-227."""
    d = (dist.unsqueeze(-1) - 0.0) / (max_val - 0.0)
    x = d.expand(-1, dim_)
    mean = P[f"{prefix}.param_module.mean"] + 0.0
    std = torch.nn.functional.softplus(P[f"{prefix}.param_module.std_logit"]) + 1e-5
    weight = torch.sigmoid(P[f"{prefix}.param_module.weight_logit"]) * (4.0 * math.sqrt(dim_))
    return torch.exp(-0.5 * (((x - mean) / std) ** 2)) * weight


# =================================================================================================
# graph construction (torch_cluster.radius restated) and edge encoding
# =================================================================================================

def radius_bipartite(x_src: Tensor, x_dst: Tensor, r: float, max_neighbors: int = 1000):
    """torch_cluster.radius(x=src, y=dst, r) as used at graph_parser.py:339: all pairs with |x-y| < r,
    at most `max_neighbors` per dst (sources scanned in index order), grouped by dst.  Done in float64-free
    plain arithmetic on the input dtype, chunked over dst to bound memory."""
    dst_idx, src_idx = [], []
    r2 = r * r
    CH = 4096
    for s in range(0, x_dst.shape[0], CH):
        y = x_dst[s:s + CH]
        d2 = ((y[:, None, :] - x_src[None, :, :]) ** 2).sum(-1)
        m = d2 < r2
        if max_neighbors < x_src.shape[0]:
            m = m & (torch.cumsum(m.to(torch.int64), dim=1) <= max_neighbors)
        di, si = m.nonzero(as_tuple=True)
        dst_idx.append(di + s)
        src_idx.append(si)
    return torch.cat(dst_idx), torch.cat(src_idx)


class GraphEdge(NamedTuple):
    edge_src: Tensor
    edge_dst: Tensor
    edge_length: Tensor
    edge_attr: Tensor
    edge_scalars: Tensor
    edge_logits: Optional[Tensor]


def encode_edges(x_src, x_dst, edge_src, edge_dst, *, r_cutoff: Optional[float], r_mincut: Optional[float],
                 irreps_sh: Irreps, length_enc, fill_edge_weights: Optional[float],
                 cutoff_eps: float = 1e-12) -> GraphEdge:
    """GraphEdgeEncoderBase._encode_edges — graph_parser.py:146-224 (sh_cutoff=False branch, offset None)."""
    edge_vec = x_src.index_select(0, edge_src) - x_dst.index_select(0, edge_dst)
    edge_length = edge_vec.norm(dim=1, p=2)
    if r_cutoff is None:
        edge_cutoff = None
    else:
        edge_cutoff = soft_square_cutoff_2(edge_length, (None, None, 0.8 * r_cutoff, 1.0 * r_cutoff))
    if r_mincut is None:
        cutoff_nonscalar = None
    else:
        cutoff_nonscalar = soft_square_cutoff_2(edge_length, (0.2 * r_mincut, 1.0 * r_mincut, None, None))
    edge_scalars = length_enc(edge_length)
    edge_sh = spherical_harmonics(irreps_sh, edge_vec)
    if cutoff_nonscalar is not None:                        # irreps_utils.cutoff_irreps :20-63
        parts, s = [], 0
        for mul, l in irreps_sh:
            d = mul * (2 * l + 1)
            blk = edge_sh[:, s:s + d]
            parts.append(blk * cutoff_nonscalar[:, None] if l != 0 else blk)
            s += d
        edge_sh = torch.cat(parts, dim=-1)
    if edge_cutoff is None:
        log_cut = None if fill_edge_weights is None else torch.ones_like(edge_length) * math.log(fill_edge_weights)
    else:
        edge_cutoff = torch.max(edge_cutoff, torch.tensor(cutoff_eps, dtype=edge_cutoff.dtype))
        log_cut = torch.log(edge_cutoff)
    return GraphEdge(edge_src, edge_dst, edge_length, edge_sh, edge_scalars, log_cut)


# =================================================================================================
# the model
# =================================================================================================

class Config(NamedTuple):
    irreps: Irreps                 # key field output == key input == query irreps (all configs)
    irreps_sh: Irreps
    num_heads: int
    fc_neurons: List[int]          # resolved, e.g. [128, 128, 64]
    length_emb_dim: int
    r_cluster_multiscale: List[Optional[float]]
    r_mincut_nonscalar_sh: float
    length_enc_max_r: Optional[float]
    time_emb_mlp: List[int]
    max_time: float
    time_enc_n: float
    lin_mult: float
    ang_mult: float
    irreps_mlp_mid: int = 3
    max_neighbors: int = 1000
    edge_time_encoding: bool = True
    use_src_point_attn: bool = False   # PointAttentiveScoreModel (point_attentive_score_model.py:71-72): alpha *= w_src after the softmax
    query_time_encoding: bool = False  # score_head.py:168-173: the query points carry query_time_mlp(time) as the field's destination feature


def config_from_kwargs(score_head_kwargs: dict) -> Config:
    """Same dict the reference splats into ScoreModelHead (score_head.py:32-41) after
    multiscale_score_model.py:79-85 has injected irreps_input / irreps_query_edf."""
    k = score_head_kwargs
    tf = k['key_tensor_field_kwargs']
    irreps = parse_irreps(tf['irreps_output'])
    assert parse_irreps(tf.get('irreps_input', tf['irreps_output'])) == irreps
    assert parse_irreps(k.get('irreps_query_edf', tf['irreps_output'])) == irreps
    ete = bool(k.get('edge_time_encoding', False))
    qte = bool(k.get('query_time_encoding', True))
    assert ete or qte or k.get('ebm', False), "No time encoding! Are you sure?"       # score_head.py:72-73 (the EBM head allows it)
    assert not (qte and k.get('ebm', False))
    assert tf.get('n_layers', 1) == 1 and tf.get('cutoff_method', 'edge_attn') == 'edge_attn'
    if tf.get('use_dst_point_attn', False):
        raise NotImplementedError                                  # gnn_block.py:196-197
    fc = list(tf['fc_neurons'])
    if fc[0] == -1:                                               # multiscale_tensor_field.py:63-67
        fc[0] = tf['length_emb_dim'] + (k['time_emb_mlp'][-1] if ete else 0)
    r0 = tf['r_cluster_multiscale'][0]
    rmin = tf.get('r_mincut_nonscalar_sh', None)
    if rmin is None:
        rmin = 0.01 * r0                                          # multiscale_tensor_field.py:94-96
    return Config(irreps=irreps, irreps_sh=parse_irreps(tf['irreps_sh']), num_heads=tf['num_heads'],
                  fc_neurons=fc, length_emb_dim=tf['length_emb_dim'],
                  r_cluster_multiscale=list(tf['r_cluster_multiscale']), r_mincut_nonscalar_sh=float(rmin),
                  length_enc_max_r=tf.get('length_enc_max_r', None), time_emb_mlp=list(k['time_emb_mlp']),
                  max_time=float(k['max_time']), time_enc_n=float(k.get('time_enc_n', 10000.)),
                  lin_mult=float(k['lin_mult']), ang_mult=float(k['ang_mult']),
                  irreps_mlp_mid=tf.get('irreps_mlp_mid', 3), edge_time_encoding=ete,
                  use_src_point_attn=bool(tf.get('use_src_point_attn', False)), query_time_encoding=qte)


class FeaturedPoints(NamedTuple):
    x: Tensor
    f: Tensor
    b: Tensor
    w: Optional[Tensor] = None


def separable_fctp_dtp_lin(cfg_in1: Irreps, cfg_in2: Irreps, ir_out: Irreps, use_activation: bool):
    """Irreps bookkeeping of SeparableFCTP.__init__ — graph_attention_transformer.py:71-116."""
    dtp = depthwise_tp(cfg_in1, cfg_in2, ir_out)
    scalars, gates, gated = irreps2gate(ir_out)
    lin_out = simplify(scalars + gates + gated) if use_activation else ir_out
    return dtp, simplify(dtp.irout), lin_out, (scalars, gates, gated)


class Debug(dict):
    """optional capture of intermediates for kernel-stage parity tests"""


def time_embeddings(cfg: Config, P, time: Tensor) -> List[Tensor]:
    """score_head.py:159-164 (per-scale MLP on the sinusoidal encoding) -> (nT, 64) per scale."""
    enc = sinusoidal_embedding(time, cfg.time_emb_mlp[0], cfg.max_time, cfg.time_enc_n)
    outs = []
    for n in range(len(cfg.r_cluster_multiscale)):
        x = enc
        li = 0
        for i in range(1, len(cfg.time_emb_mlp)):
            x = x @ P[f"time_mlps_multiscale.{n}.{li}.weight"].t() + P[f"time_mlps_multiscale.{n}.{li}.bias"]
            li += 1
            if i != len(cfg.time_emb_mlp) - 1:
                x = torch.nn.functional.silu(x)
                li += 1
        outs.append(x)
    return outs


def query_time_embedding(cfg: Config, P, time: Tensor) -> Tensor:
    """score_head.py:64-70, 171: query_time_mlp (Linear, SiLU, ..., Linear) on the sinusoidal encoding -> (nT, time_emb_mlp[-1])."""
    x = sinusoidal_embedding(time, cfg.time_emb_mlp[0], cfg.max_time, cfg.time_enc_n)
    li = 0
    for i in range(1, len(cfg.time_emb_mlp)):
        x = x @ P[f"query_time_mlp.{li}.weight"].t() + P[f"query_time_mlp.{li}.bias"]
        li += 1
        if i != len(cfg.time_emb_mlp) - 1:
            x = torch.nn.functional.silu(x)
            li += 1
    return x


def equiformer_block(cfg: Config, P, blk: str, src_f: Tensor, edge_src: Tensor, edge_dst: Tensor, edge_attr: Tensor, edge_scalars: Tensor,
                     edge_logits: Tensor, N_dst: int, src_w: Optional[Tensor] = None, irreps_output: Optional[Irreps] = None,
                     dst_f: Optional[Tensor] = None, irreps_dst: Optional[Irreps] = None):
    """EquiformerBlock.forward (gnn_block.py:164-218) + GraphAttentionMLP2.forward (graph_attention.py:218-273) on a given bipartite graph: source
    features, edge lists and the per-edge attributes / scalars / pre-attention logits.  ``blk``: the block's prefix in the state dict.
    ``dst_f`` / ``irreps_dst``: the destination features of a ``use_dst_feature=True`` block (gnn_block.py:126-130: LayerNorm + LinearRS with bias on
    them joins the message, linear_src loses its bias; :111: their projection skip_1 joins the attention output).  Returns (output features,
    intermediates)."""
    irreps, irreps_sh, H = cfg.irreps, cfg.irreps_sh, cfg.num_heads
    msg_src = equivariant_layer_norm_v2(src_f, irreps, P, f"{blk}.prenorm_src")
    msg_dst = None
    if dst_f is None:      # use_dst_feature=False: no dst message, skip_1=None
        msg_src = linear_rs(msg_src, irreps, irreps, P, f"{blk}.linear_src", bias=True)
        message = msg_src[edge_src]
    else:
        msg_src = linear_rs(msg_src, irreps, irreps, P, f"{blk}.linear_src", bias=False)
        msg_dst = equivariant_layer_norm_v2(dst_f, irreps_dst, P, f"{blk}.prenorm_dst")
        msg_dst = linear_rs(msg_dst, irreps_dst, irreps, P, f"{blk}.linear_dst", bias=True)
        message = msg_src[edge_src] + msg_dst[edge_dst]

    # ---- GraphAttentionMLP2 ------------------------------------------------------------------------
    ga = f"{blk}.ga"
    irreps_head = [(m // H, l) for m, l in irreps]
    mul_alpha = irreps[0][0]
    assert irreps[0][1] == 0
    dtp1, dtp1_out_simpl, lin1_out, gate1 = separable_fctp_dtp_lin(irreps, irreps_sh, irreps, True)
    weight = radial_profile(edge_scalars, P, f"{ga}.sep_act.dtp_rad", len(cfg.fc_neurons))
    m1 = dtp1(message, edge_attr, weight)                                   # (E, 1568)
    log_alpha = linear_rs(m1, dtp1.irout, [(mul_alpha, 0)], P, f"{ga}.sep_alpha")   # un-simplified input irreps
    log_alpha = vec2heads(log_alpha, [(mul_alpha // H, 0)], H)              # (E, H, 16)
    value = linear_rs(m1, dtp1_out_simpl, lin1_out, P, f"{ga}.sep_act.lin")
    value = gate(value, *gate1)                                             # (E, 240)
    dtp2, dtp2_out_simpl, lin2_out, _ = separable_fctp_dtp_lin(irreps, irreps_sh, irreps, False)
    v2 = dtp2(value, edge_attr, P[f"{ga}.sep_value.dtp.tp.weight"])
    value = linear_rs(v2, dtp2_out_simpl, lin2_out, P, f"{ga}.sep_value.lin")
    log_alpha = smooth_leaky_relu_n(log_alpha)
    log_alpha = torch.einsum('ehk,hk->eh', log_alpha, P[f"{ga}.alpha_dot"].squeeze(0))
    log_alpha = log_alpha + edge_logits.unsqueeze(-1)
    value = vec2heads(value, irreps_head, H)                                # (E, H, 60)
    # scatter_logsumexp / scatter-sum over dst (torch_scatter restated; empty segments -> 0)
    mx = torch.full((N_dst, H), -float('inf'), dtype=log_alpha.dtype)
    mx = mx.scatter_reduce(0, edge_dst[:, None].expand(-1, H), log_alpha, reduce='amax', include_self=True)
    mx_safe = torch.where(torch.isinf(mx), torch.zeros_like(mx), mx)
    ssum = torch.zeros((N_dst, H), dtype=log_alpha.dtype).index_add_(0, edge_dst, torch.exp(log_alpha - mx_safe[edge_dst]))
    log_Z = torch.log(ssum) + mx_safe
    alpha = torch.exp(log_alpha - log_Z[edge_dst])
    if cfg.use_src_point_attn:                                   # gnn_block.py:190-194, graph_attention.py:257-258: after the softmax
        assert isinstance(src_w, Tensor)
        alpha = alpha * src_w[edge_src].unsqueeze(-1)
    attn = value * alpha.unsqueeze(-1)
    attn = torch.zeros((N_dst,) + attn.shape[1:], dtype=attn.dtype).index_add_(0, edge_dst, attn)
    attn = heads2vec(attn, irreps_head)
    emb = linear_rs(attn, irreps, irreps, P, f"{ga}.proj")
    if dst_f is not None:                                         # skip_1 = ProjectIfMismatch(irreps_dst -> irreps_emb, layernorm=False), gnn_block.py:111, 205-206
        if list(irreps_dst) == list(irreps):
            emb = emb + dst_f
        else:
            emb = emb + linear_rs(dst_f, irreps_dst, irreps, P, f"{blk}.skip_1.skip", bias=True)

    # ---- post-norm + FFN + skip_2 (Identity) ----------------------------------------------------------
    out = equivariant_layer_norm_v2(emb, irreps, P, f"{blk}.post_norm")
    mid = simplify(sort_even_first([(m, l) for _ in range(cfg.irreps_mlp_mid) for m, l in irreps])[0])
    sc, gt, gd = irreps2gate(mid)
    ffn_in = simplify(sc + gt + gd)
    y1 = torch.ones_like(out[:, 0:1])
    t1 = fctp(irreps, [(1, 0)], ffn_in)
    h = t1(out, y1, P[f"{blk}.ffn.fctp_1.tp.weight"])
    h = add_bias(h, ffn_in, P, f"{blk}.ffn.fctp_1")
    h = gate(h, sc, gt, gd)
    ir_out = irreps if irreps_output is None else irreps_output
    t2 = fctp(mid, [(1, 0)], ir_out)
    o = t2(h, y1, P[f"{blk}.ffn.fctp_2.tp.weight"])
    o = add_bias(o, ir_out, P, f"{blk}.ffn.fctp_2")
    if list(ir_out) == list(irreps):
        o = o + emb                                               # skip_2 = Identity
    else:
        o = o + linear_rs(emb, irreps, ir_out, P, f"{blk}.skip_2.skip", bias=True)      # ProjectIfMismatch(layernorm=False)
    return o, dict(msg_src=msg_src, msg_dst=msg_dst, dtp_weight=weight, log_alpha=log_alpha, value=value, attn=attn, emb=emb)


def key_tensor_field(cfg: Config, P, query_x: Tensor, key_pcd_multiscale: Sequence[FeaturedPoints],
                     context_emb: List[Tensor], dbg: Optional[Debug] = None, pre: str = "key_tensor_field",
                     irreps_output: Optional[Irreps] = None, query_f: Optional[Tensor] = None, irreps_query: Optional[Irreps] = None) -> Tensor:
    """MultiscaleTensorField.forward (multiscale_tensor_field.py:192-260) + EquiformerBlock.forward
    (gnn_block.py:164-218) + GraphAttentionMLP2.forward (graph_attention.py:218-273).
    ``pre``: where the module sits in the state dict (``tensor_field`` / ``weight_field`` inside a KeypointExtractor);
    ``irreps_output``: the field's output irreps when they differ from its input irreps (the KeypointExtractor's weight field,
    keypoint_extractor.py:111-112): the FFN then ends in them and skip_2 is a LinearRS with bias (gnn_block.py:112);
    ``query_f`` / ``irreps_query``: the query points' own features when the field is built with ``irreps_query`` (multiscale_tensor_field.py:49-51,
    150-162: the block then runs with use_dst_feature=True)."""
    irreps, irreps_sh, H = cfg.irreps, cfg.irreps_sh, cfg.num_heads
    n_total = 0
    E_src, E_dst, E_attr, E_scal, E_logit, E_len = [], [], [], [], [], []
    fill = None
    for n, r in enumerate(cfg.r_cluster_multiscale):
        kp = key_pcd_multiscale[n]
        if r is None:
            Ns, Nd = kp.x.shape[0], query_x.shape[0]
            es = torch.arange(Ns).repeat_interleave(Nd)        # meshgrid 'ij' flattened: src-major
            ed = torch.arange(Nd).repeat(Ns)
            enc = lambda d: sinusoidal_embedding(d, cfg.length_emb_dim, float(cfg.length_enc_max_r), 1000.)
            ge = encode_edges(kp.x, query_x, es, ed, r_cutoff=None, r_mincut=cfg.r_mincut_nonscalar_sh,
                              irreps_sh=irreps_sh, length_enc=enc, fill_edge_weights=fill)
        else:
            ed, es = radius_bipartite(kp.x, query_x, float(r), cfg.max_neighbors)
            enc = lambda d, n=n, r=r: gaussian_radial_basis(d, P, f"{pre}.graph_parsers.{n}.length_enc",
                                                            cfg.length_emb_dim, float(r))
            ge = encode_edges(kp.x, query_x, es, ed, r_cutoff=float(r), r_mincut=cfg.r_mincut_nonscalar_sh,
                              irreps_sh=irreps_sh, length_enc=enc, fill_edge_weights=None)
            fill = 1.0                                           # multiscale_tensor_field.py:139-140
        scal = ge.edge_scalars
        if context_emb is not None:                               # multiscale_tensor_field.py:219-231
            scal = torch.cat([scal, context_emb[n].index_select(0, ge.edge_dst)], dim=-1)
        scal = scal @ P[f"{pre}.edge_scalars_pre_linears.{n}.0.weight"].t() + P[f"{pre}.edge_scalars_pre_linears.{n}.0.bias"]
        scal = torch.nn.functional.silu(scal)
        E_src.append(ge.edge_src + n_total)
        E_dst.append(ge.edge_dst)
        E_attr.append(ge.edge_attr)
        E_scal.append(scal)
        E_len.append(ge.edge_length)
        E_logit.append(ge.edge_logits if ge.edge_logits is not None else torch.zeros_like(ge.edge_length))
        n_total += kp.x.shape[0]
    edge_src, edge_dst = torch.cat(E_src), torch.cat(E_dst)
    edge_attr, edge_scalars, edge_logits = torch.cat(E_attr), torch.cat(E_scal), torch.cat(E_logit)
    src_f = torch.cat([kp.f for kp in key_pcd_multiscale], dim=0)
    N_dst = query_x.shape[0]

    src_w = None
    if cfg.use_src_point_attn:                                   # gnn_block.py:190-194: the key points' weights
        for kp in key_pcd_multiscale:
            assert isinstance(kp.w, Tensor)
        src_w = torch.cat([kp.w for kp in key_pcd_multiscale], dim=0)
    o, mid_ = equiformer_block(cfg, P, f"{pre}.gnn_block_init", src_f, edge_src, edge_dst, edge_attr, edge_scalars, edge_logits, N_dst, src_w, irreps_output,
                               dst_f=query_f, irreps_dst=irreps_query)
    msg_src, weight, log_alpha, value, attn, emb = (mid_[k] for k in ("msg_src", "dtp_weight", "log_alpha", "value", "attn", "emb"))
    if dbg is not None:
        dbg.update(edge_src=edge_src, edge_dst=edge_dst, edge_attr=edge_attr, edge_scalars=edge_scalars,
                   edge_logits=edge_logits, edge_length=torch.cat(E_len), msg_src=msg_src, dtp_weight=weight,
                   log_alpha=log_alpha, value=value, attn=attn, emb=emb, field=o, msg_dst=mid_["msg_dst"],
                   n_edges_per_scale=[len(e) for e in E_src])
    return o


def score_head_forward(cfg: Config, P, Ts: Tensor, key_pcd_multiscale: Sequence[FeaturedPoints],
                       query_pcd: FeaturedPoints, time: Tensor, dbg: Optional[Debug] = None):
    """ScoreModelHead.forward — score_head.py:142-211.  Returns (ang_vel, lin_vel), each (nT, 3)."""
    assert Ts.ndim == 2 and Ts.shape[-1] == 7
    assert time.ndim == 1 and len(time) == len(Ts)
    irreps = cfg.irreps
    nT, nQ = len(Ts), len(query_pcd.x)
    tembs = None
    if cfg.edge_time_encoding:
        tembs = [t.unsqueeze(-2).expand(-1, nQ, -1).reshape(nT * nQ, -1) for t in time_embeddings(cfg, P, time)]
    f_t = transform_feature_quaternion(irreps, query_pcd.f, Ts[..., :4])        # (nT, nQ, F)
    x_t = transform_points(query_pcd.x, Ts)                                     # (nT, nQ, 3)
    qf = f_t.clone().reshape(nT * nQ, -1)
    query_f, irreps_query = None, None
    if cfg.query_time_encoding:                                                 # score_head.py:168-173
        te = cfg.time_emb_mlp[-1]
        query_f = query_time_embedding(cfg, P, time).unsqueeze(-2).expand(nT, nQ, te).reshape(nT * nQ, te)
        irreps_query = [(te, 0)]                                                # score_head.py:52, 81-83
    field = key_tensor_field(cfg, P, x_t.reshape(-1, 3), key_pcd_multiscale, tembs, dbg, query_f=query_f, irreps_query=irreps_query)

    n_pre = sum(m for m, l in irreps if l == 1)           # (query 1e + key 1e)//2 with equal irreps
    ir_out = [(1, 0), (n_pre, 1)]
    outs = []
    for name in ("lin_vel_tp", "ang_vel_tp"):
        dtp, dtp_simpl, lin_out, gts = separable_fctp_dtp_lin(irreps, irreps, ir_out, True)
        t = dtp(qf, field, P[f"{name}.dtp.tp.weight"])
        t = linear_rs(t, dtp_simpl, lin_out, P, f"{name}.lin")
        t = gate(t, *gts)
        outs.append(t[..., 1:].reshape(nT, nQ, n_pre, 3).mean(dim=-2))
    lin_vel, ang_spin = outs
    q = Ts[..., :4]
    qinv = quaternion_invert(q.unsqueeze(-2))
    lin_vel = quaternion_apply(qinv, lin_vel)
    ang_spin = quaternion_apply(qinv, ang_spin)
    ang_orbital = torch.cross(query_pcd.x.unsqueeze(0) / cfg.lin_mult, lin_vel, dim=-1)
    w = query_pcd.w
    if dbg is not None:
        dbg.update(x_t=x_t, f_t=f_t, lin_vel_q=lin_vel, ang_spin_q=ang_spin)
    lin = torch.einsum('q,tqi->ti', w, lin_vel)
    ang = torch.einsum('q,tqi->ti', w, ang_orbital) + torch.einsum('q,tqi->ti', w, ang_spin)
    return ang, lin


def compute_energy(cfg: Config, P, Ts: Tensor, key_pcd_multiscale: Sequence[FeaturedPoints], query_pcd: FeaturedPoints,
                   time: Tensor, dbg: Optional[Debug] = None) -> Tensor:
    """EbmScoreModelHead.compute_energy — score_head_ebm.py:122-174 (the critic used by agent.py:163-174 to rank poses).
    energy_t = sum_q w_q |field(T_t x_q) - D(q_t) f_q|^2 / dim."""
    assert Ts.ndim == 2 and Ts.shape[-1] == 7
    irreps = cfg.irreps
    nT, nQ = len(Ts), len(query_pcd.x)
    tembs = None
    if cfg.edge_time_encoding:
        tembs = [t.unsqueeze(-2).expand(-1, nQ, -1).reshape(nT * nQ, -1) for t in time_embeddings(cfg, P, time)]
    f_t = transform_feature_quaternion(irreps, query_pcd.f, Ts[..., :4])
    x_t = transform_points(query_pcd.x, Ts)
    field = key_tensor_field(cfg, P, x_t.reshape(-1, 3), key_pcd_multiscale, tembs, dbg)
    qf = f_t.reshape(nT * nQ, -1)
    energy = (field - qf).square().sum(dim=-1) * (1.0 / float(dim(irreps)))
    return torch.einsum('q,tq->t', query_pcd.w, energy.view(nT, nQ))


_Q_INDICES = torch.tensor([[1, 2, 3], [0, 3, 2], [3, 0, 1], [2, 1, 0]], dtype=torch.long)
_Q_FACTOR = torch.tensor([[-0.5, -0.5, -0.5], [0.5, -0.5, 0.5], [0.5, 0.5, -0.5], [-0.5, 0.5, 0.5]], dtype=torch.float64)


def t_schedule(schedule: Tuple[float, float], n_steps: int, log_t: bool = True) -> Tensor:
    """score_model_base.py:146-164."""
    sch = torch.tensor(schedule, dtype=torch.float64)
    if log_t:
        return torch.logspace(start=torch.log(sch[0]), end=torch.log(sch[1]), steps=n_steps, base=torch.e,
                              dtype=torch.float64)
    return torch.linspace(start=sch[0], end=sch[1], steps=n_steps, dtype=torch.float64)


def langevin_step(cfg: Config, T: Tensor, ang_dimless: Tensor, lin_dimless: Tensor, t: float, dt: float,
                  temperature_base: float, time_exponent_temp: float, time_exponent_alpha: float,
                  noise_ang: Tensor, noise_lin: Tensor) -> Tensor:
    """One iteration of ScoreModelBase.sample's inner loop — score_model_base.py:168-193 (float64)."""
    t = torch.tensor(t, dtype=torch.float64)
    temperature = temperature_base * torch.pow(t, time_exponent_temp)
    alpha_ang = (cfg.ang_mult ** 2) * torch.pow(t, time_exponent_alpha) * dt
    alpha_lin = (cfg.lin_mult ** 2) * torch.pow(t, time_exponent_alpha) * dt
    ang_score = ang_dimless.double() / (cfg.ang_mult * torch.sqrt(t))
    lin_score = lin_dimless.double() / (cfg.lin_mult * torch.sqrt(t))
    ang_disp = (alpha_ang / 2) * ang_score + torch.sqrt(temperature * alpha_ang) * noise_ang
    lin_disp = (alpha_lin / 2) * lin_score + torch.sqrt(temperature * alpha_lin) * noise_lin
    L = T[..., _Q_INDICES] * _Q_FACTOR
    q, x = T[..., :4], T[..., 4:]
    dq = torch.einsum('...ij,...j->...i', L, ang_disp)
    dx = quaternion_apply(q, lin_disp)
    q = q + dq
    q = q / torch.norm(q, dim=-1, keepdim=True)
    return torch.cat([q, x + dx], dim=-1)


def sample(cfg: Config, P, T_seed: Tensor, key_pcd_multiscale, query_pcd, diffusion_schedules, N_steps,
           timesteps, temperatures=1.0, log_t_schedule=True, time_exponent_temp=0.5, time_exponent_alpha=0.5,
           noise: Optional[Tensor] = None, compute_dtype=torch.float32) -> Tensor:
    """ScoreModelBase.sample — score_model_base.py:110-204.  `noise`: optional (sum N_steps, 2, nT, 3)
    float64 standard normals [ang, lin] replacing torch.randn_like (for parity runs)."""
    if isinstance(temperatures, (int, float)):
        temperatures = [float(temperatures)] * len(diffusion_schedules)
    T = T_seed.clone().double()
    Ts = [T.clone()]
    step = 0
    for n, sch in enumerate(diffusion_schedules):
        ts = t_schedule(sch, N_steps[n], log_t_schedule)
        for i in range(len(ts)):
            t = ts[i]
            ang, lin = score_head_forward(cfg, P, T.to(compute_dtype), key_pcd_multiscale, query_pcd,
                                          t.repeat(len(T)).to(compute_dtype))
            if noise is None:
                na, nl = torch.randn_like(ang, dtype=torch.float64), torch.randn_like(lin, dtype=torch.float64)
            else:
                na, nl = noise[step, 0], noise[step, 1]
            T = langevin_step(cfg, T, ang, lin, float(t), timesteps[n], temperatures[n], time_exponent_temp,
                              time_exponent_alpha, na, nl)
            step += 1
            Ts.append(T.clone())
    Ts.append(T.clone())
    return torch.stack(Ts, dim=0)


def agent_sample(models, critic, T0: Tensor, N_steps_list, timesteps_list, temperatures_list, diffusion_schedules_list,
                 noise_list=None, log_t_schedule=True, time_exponent_temp=1.0, time_exponent_alpha=0.5,
                 compute_dtype=torch.float32):
    """DiffusionEdfAgent.sample — agent.py:98-186 on already extracted features: the models denoise one after the other,
    each starting from the previous one's final poses (:139-156); the trajectories are concatenated (:157) and, with a critic,
    the poses are reordered by ascending energy of the final poses (:159-174).
    `models`: [(cfg, P, key_pcd_multiscale, query_pcd)], `critic`: the same tuple or None.  Returns (Ts_out, energy_sorted)."""
    assert len(models) == len(N_steps_list) == len(timesteps_list) == len(temperatures_list) == len(diffusion_schedules_list)
    if noise_list is None:
        noise_list = [None] * len(models)
    outs = []
    for (cfg, P, keys, query), N_steps, timesteps, temperatures, sched, noise in zip(
            models, N_steps_list, timesteps_list, temperatures_list, diffusion_schedules_list, noise_list):
        assert len(sched) == len(N_steps) and len(sched) == len(timesteps)
        Ts = sample(cfg, P, T0.clone(), keys, query, sched, N_steps, timesteps, temperatures, log_t_schedule,
                    time_exponent_temp, time_exponent_alpha, noise=noise, compute_dtype=compute_dtype)
        T0 = Ts[-1]
        outs.append(Ts)
    Ts_out = torch.cat(outs, dim=0)
    energy_sorted = None
    if critic is not None:
        cfg, P, keys, query = critic
        T_last = Ts_out[-1]
        energy = compute_energy(cfg, P, T_last.to(P[next(iter(P))].dtype), keys, query,
                                torch.ones(len(T_last), dtype=P[next(iter(P))].dtype))
        energy_sorted, idx = energy.sort(descending=False)
        Ts_out = Ts_out[..., idx, :]
    return Ts_out, energy_sorted


def cast_params(P: Dict[str, Tensor], dtype) -> Dict[str, Tensor]:
    return {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in P.items()}
