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
    irreps: R.Irreps            # irreps_src == irreps_dst == irreps_emb of the layer
    irreps_sh: R.Irreps
    num_heads: int
    fc_neurons: list            # [num_basis, h1, h2]
    radius: float               # the level's connection radius; the radial basis uses cutoff = 0.99 * radius
    irreps_mlp_mid: int = 3


def soft_step(x, n: int = 3):                        # radial_func.py:15-17
    return (x > 0) * ((x < 1) * ((n + 1) * x.pow(n) - n * x.pow(n + 1)) + (x >= 1))


def soft_cutoff(x, thr: float = 0.8, n: int = 3):    # radial_func.py:19-22
    return 1 - soft_step((x - thr) / (1 - thr), n=n)


def soft_square_cutoff(x, thr: float = 0.8, n: int = 3, infinite: bool = False):      # radial_func.py:24-29
    if infinite:
        return soft_cutoff(x, thr=thr, n=n) * (x > 0.5) + soft_cutoff(1 - x, thr=thr, n=n) * (x <= 0.5)
    return (x > 0.5) + soft_cutoff(1 - x, thr=thr, n=n) * (x <= 0.5)


def radial_basis_finite_cutoff(dist: Tensor, mean: Tensor, std_logit: Tensor, weight_logit: Tensor, cutoff: float,
                               offset: Optional[float] = None, cutoff_thr_ratio: float = 0.8) -> Tensor:
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
    return x * math.sqrt(num_basis)


def layer_forward(cfg: LayerConfig, P: Dict[str, Tensor], x_src: Tensor, f_src: Tensor, x_dst: Tensor, f_dst: Tensor,
                  edge_src: Tensor, edge_dst: Tensor, dbg: Optional[dict] = None) -> Tensor:
    """one UNet layer: radial basis + block.EquiformerBlock.forward (block.py:141-174) -> new destination features"""
    irreps, irreps_sh, H = cfg.irreps, cfg.irreps_sh, cfg.num_heads
    N_dst = x_dst.shape[0]
    edge_vec = x_src.index_select(0, edge_src) - x_dst.index_select(0, edge_dst)          # unet_feature_extractor.py:289
    edge_length = edge_vec.norm(dim=1, p=2)
    edge_attr = R.spherical_harmonics(irreps_sh, edge_vec)
    edge_scalars = radial_basis_finite_cutoff(edge_length, P["radial.mean"], P["radial.std_logit"], P["radial.weight_logit"],
                                              cutoff=0.99 * cfg.radius)
    g = "gnn"
    # block.py:149-153: the LayerNorm outputs are overwritten -> the linears act on the raw inputs
    msg_src = R.linear_rs(f_src, irreps, irreps, P, f"{g}.linear_src", bias=False)
    msg_dst = R.linear_rs(f_dst, irreps, irreps, P, f"{g}.linear_dst", bias=True)
    message = msg_src[edge_src] + msg_dst[edge_dst]

    ga = f"{g}.ga"                                                                         # graph_attention.py:84-122
    irreps_head = [(m // H, l) for m, l in irreps]
    mul_alpha = irreps[0][0]
    dtp1, dtp1_out_simpl, lin1_out, gate1 = R.separable_fctp_dtp_lin(irreps, irreps_sh, irreps, True)
    weight = R.radial_profile(edge_scalars, P, f"{ga}.sep_act.dtp_rad", len(cfg.fc_neurons))
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
    nf = R.equivariant_layer_norm_v2(node_output, irreps, P, f"{g}.norm_2")
    mid = R.simplify(R.sort_even_first([(m, l) for _ in range(cfg.irreps_mlp_mid) for m, l in irreps])[0])
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
