"""CPU restatement of ONE layer of the reference's UNet feature extractor — TEST INFRASTRUCTURE (the oracle of SURVEY §8(f) row 1's
first slice; imported by tests/ only, never by the product).

A "layer" is the ModuleDict pair the extractor builds everywhere (reference ``unet_feature_extractor.py:141-156, 160-176, 187-202``):

    layer['radial'] = GaussianRadialBasisLayerFiniteCutoff(num_basis = fc_neurons[0], cutoff = 0.99 r)     radial_func.py:231-278
    layer['gnn']    = block.EquiformerBlock(irreps_src, irreps_dst, irreps_edge_attr, irreps_head, num_heads, fc_neurons,
                                            irreps_mlp_mid = 3, attn_type = 'mlp', src_bias = False, dst_bias = True)   block.py:62-174

applied to a bipartite graph (edge_src -> edge_dst) as in ``unet_feature_extractor.py:289-302`` (pool layer) / ``:316-324``
(radius-graph layers):  edge_vec = x_src[edge_src] - x_dst[edge_dst], SH(normalize=True, 'component'), radial basis of the length.

Quirk restated deliberately (``block.py:149-153``): the results of ``norm_1_src`` / ``norm_1_dst`` are overwritten — the linear
layers see the UN-normalised inputs; the two norms' parameters exist in the state dict and have no effect.

``GraphAttentionMLP`` (``graph_attention.py:11-122``) differs from the score head's ``GraphAttentionMLP2`` only by its inputs: the
message is ``linear_src(f_src)[edge_src] + linear_dst(f_dst)[edge_dst]``, the radial MLP reads the radial basis directly (no
pre-linear), there is no additive edge logit; everything else is the code path restated in ``restatement.key_tensor_field``.

Parity status: the radial basis is pinned by golden vectors of the importable reference module (tests/golden/unet_radial.npz);
the e3nn / torch_scatter semantics are those of ``restatement.py`` (parity unpinned, see its header).
"""
from __future__ import annotations

import math
from typing import Dict, NamedTuple, Optional

import torch
from torch import Tensor

from . import restatement as R


class LayerConfig(NamedTuple):
    irreps: R.Irreps            # irreps_dst == irreps_emb of the layer
    irreps_sh: R.Irreps
    num_heads: int
    fc_neurons: list            # [num_basis, h1, h2]
    radius: float               # the level's connection radius; the radial basis uses cutoff = 0.99 * radius
    irreps_mlp_mid: int = 3
    irreps_src: Optional[R.Irreps] = None      # default: same as irreps (pool / unpool layers between levels of different width differ)
    # "embedded model" switches, used ONLY to validate diffusion_edf_amd/unet_pad.py on the CPU: a narrow layer zero-padded into the wide
    # shape equals the narrow layer iff the LayerNorms use the true channel counts and the radial basis the true num_basis
    valid: Optional[list] = None               # true multiplicities per degree (norm_2 statistics)
    fc_valid: Optional[list] = None            # true [num_basis, h1, h2]
    mid_muls: Optional[list] = None            # hidden multiplicities of the FFN when they are not irreps_mlp_mid x mul (the kernels' lmax-3 shape:
                                               # 24x3e hidden channels padded to 32, not to 3 x 16)


def _masked_layer_norm(x: Tensor, n_valid: int, w: Tensor, b: Tensor, eps: float = 1e-5) -> Tensor:
    """LayerNorm whose statistics run over the n_valid true channels of a zero-padded vector (padded entries are 0 on entry)"""
    mean = x.sum(-1, keepdim=True) / n_valid
    var = ((x - mean) ** 2).sum(-1, keepdim=True) - (x.shape[-1] - n_valid) * mean ** 2
    return (x - mean) / torch.sqrt(var / n_valid + eps) * w + b


def _radial_profile(x: Tensor, P, prefix: str, fc_valid) -> Tensor:
    if fc_valid is None:
        return R.radial_profile(x, P, prefix, 3)
    idx = 0
    for i in range(3):
        x = x @ P[f"{prefix}.net.{idx}.weight"].t()
        if i < 2:
            x = x + P[f"{prefix}.net.{idx}.bias"]
            x = _masked_layer_norm(x, fc_valid[1 + i], P[f"{prefix}.net.{idx + 1}.weight"], P[f"{prefix}.net.{idx + 1}.bias"])
            x = torch.nn.functional.silu(x)
            idx += 3
    return x + P[f"{prefix}.offset"].reshape(1, -1)


def _norm_v2(x: Tensor, irreps, P, prefix: str, valid, eps: float = 1e-5) -> Tensor:
    if valid is None:
        return R.equivariant_layer_norm_v2(x, irreps, P, prefix)
    fields, ix, iw = [], 0, 0
    for (mul, l), nv in zip(irreps, valid):
        d = 2 * l + 1
        f = x[:, ix:ix + mul * d].reshape(-1, mul, d)
        ix += mul * d
        if l == 0:
            mean = f.sum(dim=1, keepdim=True) / nv
            f = f - mean
            norm = (f.pow(2).mean(-1).sum(dim=1, keepdim=True) - (mul - nv) * mean[..., 0] ** 2) / nv
        else:
            norm = f.pow(2).mean(-1).sum(dim=1, keepdim=True) / nv
        norm = (norm + eps).pow(-0.5) * P[f"{prefix}.affine_weight"][None, iw:iw + mul]
        iw += mul
        f = f * norm.reshape(-1, mul, 1)
        if d == 1:
            f = f + P[f"{prefix}.affine_bias"][:mul].reshape(mul, 1)
        fields.append(f.reshape(-1, mul * d))
    return torch.cat(fields, dim=-1)


def soft_step(x, n: int = 3):                        # radial_func.py:15-17
    return (x > 0) * ((x < 1) * ((n + 1) * x.pow(n) - n * x.pow(n + 1)) + (x >= 1))


def soft_cutoff(x, thr: float = 0.8, n: int = 3):    # radial_func.py:19-22
    return 1 - soft_step((x - thr) / (1 - thr), n=n)


def soft_square_cutoff(x, thr: float = 0.8, n: int = 3, infinite: bool = False):      # radial_func.py:24-29
    if infinite:
        return soft_cutoff(x, thr=thr, n=n) * (x > 0.5) + soft_cutoff(1 - x, thr=thr, n=n) * (x <= 0.5)
    return (x > 0.5) + soft_cutoff(1 - x, thr=thr, n=n) * (x <= 0.5)


def radial_basis_finite_cutoff(dist: Tensor, mean: Tensor, std_logit: Tensor, weight_logit: Tensor, cutoff: float,
                               offset: Optional[float] = None, cutoff_thr_ratio: float = 0.8, num_basis_norm: Optional[int] = None) -> Tensor:
    """GaussianRadialBasisLayerFiniteCutoff.forward — radial_func.py:262-278 (soft_cutoff=True, infinite=False, max_weight 4)."""
    num_basis = mean.shape[-1]
    if offset is None:
        offset = 0.01 * cutoff
    d = ((dist - offset) / (cutoff - offset)).unsqueeze(-1)
    x = d.expand(-1, num_basis)
    std = torch.nn.functional.softplus(std_logit) + 1e-5
    x = torch.exp(-0.5 * (((x - mean) / std) ** 2))
    x = torch.sigmoid(weight_logit) * 4.0 * x
    x = x * soft_square_cutoff(d, thr=cutoff_thr_ratio, infinite=False)
    return x * math.sqrt(num_basis if num_basis_norm is None else num_basis_norm)


def layer_forward(cfg: LayerConfig, P: Dict[str, Tensor], x_src: Tensor, f_src: Tensor, x_dst: Tensor, f_dst: Tensor,
                  edge_src: Tensor, edge_dst: Tensor, dbg: Optional[dict] = None) -> Tensor:
    """one UNet layer: radial basis + block.EquiformerBlock.forward (block.py:141-174) -> new destination features"""
    irreps, irreps_sh, H = cfg.irreps, cfg.irreps_sh, cfg.num_heads
    N_dst = x_dst.shape[0]
    edge_vec = x_src.index_select(0, edge_src) - x_dst.index_select(0, edge_dst)          # unet_feature_extractor.py:289
    edge_length = edge_vec.norm(dim=1, p=2)
    edge_attr = R.spherical_harmonics(irreps_sh, edge_vec)
    edge_scalars = radial_basis_finite_cutoff(edge_length, P["radial.mean"], P["radial.std_logit"], P["radial.weight_logit"],
                                              cutoff=0.99 * cfg.radius, num_basis_norm=None if cfg.fc_valid is None else cfg.fc_valid[0])
    g = "gnn"
    irreps_src = irreps if cfg.irreps_src is None else cfg.irreps_src
    # block.py:149-153: the LayerNorm outputs are overwritten -> the linears act on the raw inputs
    msg_src = R.linear_rs(f_src, irreps_src, irreps, P, f"{g}.linear_src", bias=False)
    msg_dst = R.linear_rs(f_dst, irreps, irreps, P, f"{g}.linear_dst", bias=True)
    message = msg_src[edge_src] + msg_dst[edge_dst]

    ga = f"{g}.ga"                                                                         # graph_attention.py:84-122
    irreps_head = [(m // H, l) for m, l in irreps]
    mul_alpha = irreps[0][0]
    dtp1, dtp1_out_simpl, lin1_out, gate1 = R.separable_fctp_dtp_lin(irreps, irreps_sh, irreps, True)
    weight = _radial_profile(edge_scalars, P, f"{ga}.sep_act.dtp_rad", cfg.fc_valid)
    m1 = dtp1(message, edge_attr, weight)
    log_alpha = R.linear_rs(m1, dtp1.irout, [(mul_alpha, 0)], P, f"{ga}.sep_alpha")
    log_alpha = R.vec2heads(log_alpha, [(mul_alpha // H, 0)], H)
    value = R.linear_rs(m1, dtp1_out_simpl, lin1_out, P, f"{ga}.sep_act.lin")
    value = R.gate(value, *gate1)
    dtp2, dtp2_out_simpl, lin2_out, _ = R.separable_fctp_dtp_lin(irreps, irreps_sh, irreps, False)
    v2 = dtp2(value, edge_attr, P[f"{ga}.sep_value.dtp.tp.weight"])
    value = R.linear_rs(v2, dtp2_out_simpl, lin2_out, P, f"{ga}.sep_value.lin")
    value = R.vec2heads(value, irreps_head, H)
    log_alpha = R.smooth_leaky_relu_n(log_alpha)
    log_alpha = torch.einsum('ehk,hk->eh', log_alpha, P[f"{ga}.alpha_dot"].squeeze(0))
    mx = torch.full((N_dst, H), -float('inf'), dtype=log_alpha.dtype)
    mx = mx.scatter_reduce(0, edge_dst[:, None].expand(-1, H), log_alpha, reduce='amax', include_self=True)
    mx_safe = torch.where(torch.isinf(mx), torch.zeros_like(mx), mx)
    ssum = torch.zeros((N_dst, H), dtype=log_alpha.dtype).index_add_(0, edge_dst, torch.exp(log_alpha - mx_safe[edge_dst]))
    log_Z = torch.log(ssum) + mx_safe
    alpha = torch.exp(log_alpha - log_Z[edge_dst])
    attn = value * alpha.unsqueeze(-1)
    attn = torch.zeros((N_dst,) + attn.shape[1:], dtype=attn.dtype).index_add_(0, edge_dst, attn)
    attn = R.heads2vec(attn, irreps_head)
    node_features = R.linear_rs(attn, irreps, irreps, P, f"{ga}.proj")

    node_output = f_dst + node_features                                                    # block.py:165
    nf = _norm_v2(node_output, irreps, P, f"{g}.norm_2", cfg.valid)
    mid = R.simplify(R.sort_even_first([(m, l) for _ in range(cfg.irreps_mlp_mid) for m, l in irreps])[0])
    if cfg.mid_muls is not None:
        mid = [(int(m), l) for l, m in enumerate(cfg.mid_muls)]
    sc, gt, gd = R.irreps2gate(mid)
    ffn_in = R.simplify(sc + gt + gd)
    y1 = torch.ones_like(nf[:, 0:1])
    h = R.fctp(irreps, [(1, 0)], ffn_in)(nf, y1, P[f"{g}.ffn.fctp_1.tp.weight"])
    h = R.add_bias(h, ffn_in, P, f"{g}.ffn.fctp_1")
    h = R.gate(h, sc, gt, gd)
    o = R.fctp(mid, [(1, 0)], irreps)(h, y1, P[f"{g}.ffn.fctp_2.tp.weight"])
    o = R.add_bias(o, irreps, P, f"{g}.ffn.fctp_2")
    out = node_output + o                                                                  # block.py:172
    if dbg is not None:
        dbg.update(edge_length=edge_length, edge_attr=edge_attr, edge_scalars=edge_scalars, msg_src=msg_src, msg_dst=msg_dst,
                   dtp_weight=weight, log_alpha=log_alpha, value=value, attn=attn, proj=node_features, out=out)
    return out


# =====================================================================================================================================
# The whole extractor — unet_feature_extractor.py:260-417
# =====================================================================================================================================

class UnetConfig(NamedTuple):
    irreps_input: R.Irreps
    irreps_output: R.Irreps
    irreps_emb: list            # per scale
    fc_neurons: list            # per scale
    n_layers: list
    pool_ratio: list
    radius: list                # per scale, already resolved (:78-86)
    n_layers_midstream: int = 2
    num_heads: int = 4
    irreps_sh: R.Irreps = [(1, 0), (1, 1), (1, 2)]
    max_num_neighbors: int = 1000
    output_scalespace: Optional[list] = None


def _sub(P: Dict[str, Tensor], prefix: str) -> Dict[str, Tensor]:
    return {k[len(prefix):]: v for k, v in P.items() if k.startswith(prefix)}


def project_if_mismatch(x: Tensor, ir_in, ir_out, P, prefix: str) -> Tensor:
    """skip.py:13-34: identity for equal irreps, else EquivariantLayerNormV2(irreps_in) then LinearRS(bias)"""
    if list(ir_in) == list(ir_out):
        return x
    x = R.equivariant_layer_norm_v2(x, ir_in, P, f"{prefix}.layernorm")
    return R.linear_rs(x, ir_in, ir_out, P, f"{prefix}.skip", bias=True)


def unet_forward(cfg: UnetConfig, P: Dict[str, Tensor], x: Tensor, f: Tensor, dbg: Optional[dict] = None, forward_only: bool = False):
    """-> [(coords, features)] per output scale.  ``forward_only``: ForwardOnlyFeatureExtractor (forward_only_feature_extractor.py:196-274) = the
    down path alone, one output per scale taken after its layer stack.  ``x`` (N,3) float32 — the graphs are built in float32 exactly like torch_cluster does
    (graph_oracle.py), the layers run in the dtype of ``f`` / ``P`` on those coordinates.  Single cloud (batch all 0), deterministic FPS
    (start at point 0), eval mode (dropout / drop-path are identity)."""
    from . import graph_oracle as GO
    ns, emb, dt = len(cfg.irreps_emb), cfg.irreps_emb, f.dtype
    assert x.dtype == torch.float32
    lcfg = lambda n, src, dst: LayerConfig(dst, cfg.irreps_sh, cfg.num_heads, cfg.fc_neurons[n], cfg.radius[n], irreps_src=src)
    run = lambda c, prefix, xs, fs, xd, fd, es, ed: layer_forward(c, _sub(P, prefix), xs.to(dt), fs, xd.to(dt), fd, es, ed)
    f = R.linear_rs(f, cfg.irreps_input, emb[0], P, "input_emb", bias=True)                            # :270-271
    down_out, down_edges, scale_out = [(f, x)], [], []
    scales = list(range(ns)) if cfg.output_scalespace is None else [ns + s if s < 0 else s for s in cfg.output_scalespace]
    for n in range(ns):
        prev = emb[max(n - 1, 0)]
        idx = GO.fps(x.numpy(), cfg.pool_ratio[n], start=0)                                             # FpsPool, connectivity.py:56-80
        x_dst = x[torch.from_numpy(idx)]
        ed, es = GO.radius(x.numpy(), x_dst.numpy(), cfg.radius[n], cfg.max_num_neighbors)
        keep = idx[ed] != es
        es, ed = torch.from_numpy(es[keep]), torch.from_numpy(ed[keep])
        f_dst = project_if_mismatch(f[torch.from_numpy(idx)], prev, emb[n], P, f"down_blocks.{n}.pool_proj")
        f = run(lcfg(n, prev, emb[n]), f"down_blocks.{n}.pool_layer.", x, f, x_dst, f_dst, es, ed)     # :283-299
        x = x_dst
        down_out.append((f, x)); down_edges.append((es, ed))
        ed, es = GO.radius(x.numpy(), x.numpy(), cfg.radius[n], cfg.max_num_neighbors, exclude_self=True)      # RadiusGraph, :306-313
        es, ed = torch.from_numpy(es), torch.from_numpy(ed)
        for i in range(cfg.n_layers[n] - 1):
            f = run(lcfg(n, emb[n], emb[n]), f"down_blocks.{n}.layer_stack.{i}.", x, f, x, f, es, ed)
            down_out.append((f, x)); down_edges.append((es, ed))
        scale_out.append((x, f))
    if forward_only:
        return [(scale_out[s][0], project_if_mismatch(scale_out[s][1], emb[s], cfg.irreps_output, P, f"project_outputs.{s}")) for s in range(ns) if s in scales]
    for i in range(cfg.n_layers_midstream):                                                            # :332-344
        f = run(lcfg(ns - 1, emb[-1], emb[-1]), f"mid_block.{i}.", x, f, x, f, es, ed)
    f_skip, _ = down_out.pop()
    f = (f + f_skip) / math.sqrt(3)                                                                    # :346-347
    up_out = []
    for k in range(ns):
        n = ns - 1 - k
        for i in range(cfg.n_layers[n] - 1):
            f_skip, x_dst = down_out.pop()
            es, ed = down_edges.pop()
            f_dst = (f + f_skip) / math.sqrt(3)                                                        # :359
            # :358 swaps source and destination and flips the odd harmonics; Y_l(-v) = (-1)^l Y_l(v), so that is the harmonics of the swapped
            # edge vector, which layer_forward computes itself
            f = run(lcfg(n, emb[n], emb[n]), f"up_blocks.{k}.layer_stack.{i}.", x, f, x_dst, f_dst, ed, es)
            x = x_dst
        up_out.append((x, f))
        f_dst, x_dst = down_out.pop()                                                                  # :381-383
        es, ed = down_edges.pop()
        if k != ns - 1:
            f = run(lcfg(n, emb[n], emb[max(n - 1, 0)]), f"up_blocks.{k}.unpool_layer.", x, f, x_dst, f_dst, ed, es)
            x = x_dst
    up_out = up_out[::-1]
    return [(up_out[s][0], project_if_mismatch(up_out[s][1], emb[s], cfg.irreps_output, P, f"project_outputs.{s}")) for s in range(ns) if s in scales]


# =====================================================================================================================================
# KeypointExtractor — keypoint_extractor.py:50-197
# =====================================================================================================================================

def keypoint_extractor_forward(unet_cfg: UnetConfig, field_cfg: "R.Config", P: Dict[str, Tensor], x: Tensor, f: Tensor, pool_ratio: float,
                               bbox=None, weight_sigmoid: bool = True, weight_mult: Optional[float] = None, dbg: Optional[dict] = None):
    """-> (key-point coordinates, features (nK, D), weights (nK,)).  ``x`` float32 (graphs and FPS in float32), arithmetic in the dtype of ``f``.
    UNet on the whole cloud (:172); key points = FPS over the points inside ``bbox`` (:136-150, deterministic start); ``tensor_field`` and
    ``weight_field`` (MultiscaleTensorField without context / query features, :97-112) evaluated at them; ``weight_post`` = LayerNorm, SiLU,
    Linear(·,1), Sigmoid (:113-118)."""
    from . import graph_oracle as GO
    dt = f.dtype
    scales = unet_forward(unet_cfg, _sub(P, "feature_extractor."), x, f)
    key_pcds = [R.FeaturedPoints(x=xs.to(dt), f=fs, b=torch.zeros(len(xs), dtype=torch.long), w=None) for xs, fs in scales]
    xq = x
    if bbox is not None:
        bb = torch.tensor(bbox, dtype=x.dtype)
        xq = xq[((xq >= bb[:, 0]) & (xq <= bb[:, 1])).all(dim=-1)]
    xq = xq[torch.from_numpy(GO.fps(xq.numpy(), pool_ratio, start=0))]
    feat = R.key_tensor_field(field_cfg, P, xq.to(dt), key_pcds, None, pre="tensor_field")
    wpre = R.key_tensor_field(field_cfg, P, xq.to(dt), key_pcds, None, pre="weight_field", irreps_output=[(P["weight_post.0.weight"].numel(), 0)])
    h = torch.nn.functional.layer_norm(wpre, (wpre.shape[-1],), P["weight_post.0.weight"], P["weight_post.0.bias"], 1e-5)
    w = (torch.nn.functional.silu(h) @ P["weight_post.2.weight"].t() + P["weight_post.2.bias"]).squeeze(-1)
    if weight_sigmoid:
        w = torch.sigmoid(w)
    if weight_mult is not None:
        w = w * weight_mult
    if dbg is not None:
        dbg.update(scales=scales, wpre=wpre)
    return xq, feat, w
