"""Parameter schema of the score head, name-compatible with the reference ``state_dict``.

Names follow the reference module tree (reference trainer.py:141-147 loads
``score_model_state_dict``; the score head lives under the ``score_head.`` prefix there), e.g.
``key_tensor_field.gnn_block_init.ga.sep_act.dtp_rad.net.0.weight``.  e3nn ``TensorProduct`` weights are
flat vectors in instruction order, each block shaped ``(mul_in1, mul_in2[, mul_out])``
(reference equiformer/tensor_product_rescale.py:155-173, 352-382).

``HeadConfig`` resolves the same ``score_head_kwargs`` dict the reference splats into
``ScoreModelHead.__init__`` (reference score_head.py:32-41).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from .so3 import parse_irreps, irreps_dim

Irreps = List[Tuple[int, int]]


@dataclass
class HeadConfig:
    irreps: Irreps                       # key-field output == key input == query irreps
    lmax_sh: int
    num_heads: int
    fc_neurons: List[int]                # resolved [in, h1, h2]
    length_emb_dim: int
    radii: List[Optional[float]]         # r_cluster_multiscale; None = infinite scale
    r_mincut_nonscalar_sh: float
    length_enc_max_r: float
    time_emb_mlp: List[int]
    max_time: float
    time_enc_n: float
    lin_mult: float
    ang_mult: float
    irreps_mlp_mid: int = 3
    max_neighbors: int = 1000
    ebm: bool = False                    # EbmScoreModelHead (reference score_head_ebm.py): energy critic, no time encoding
    half_gemm: bool = False              # the reference's half_precision knob (agent.py:50-51): single-term fp16 GEMM products
    use_src_point_attn: bool = False     # PointAttentiveScoreModel (point_attentive_score_model.py:71-72): attention times the key points' weights
    query_time_encoding: bool = False    # score_head.py:168-173: query_time_mlp(time) as the destination feature of the key field's block

    @property
    def n_scales(self) -> int:
        return len(self.radii)

    @property
    def muls(self) -> List[int]:
        return [m for m, _ in self.irreps]

    @property
    def lmax(self) -> int:
        return max(l for _, l in self.irreps)

    @property
    def dim(self) -> int:
        return irreps_dim(self.irreps)

    @classmethod
    def from_kwargs(cls, score_head_kwargs: dict) -> "HeadConfig":
        k = score_head_kwargs
        tf = dict(k['key_tensor_field_kwargs'])
        irreps = parse_irreps(tf['irreps_output'])
        if 'irreps_input' in tf and parse_irreps(tf['irreps_input']) != irreps:
            raise NotImplementedError("irreps_input != irreps_output is not used by any reference config")
        if 'irreps_query_edf' in k and parse_irreps(k['irreps_query_edf']) != irreps:
            raise NotImplementedError("irreps_query_edf != key irreps is not used by any reference config")
        for i, (m, l) in enumerate(irreps):
            if l != i:
                raise NotImplementedError(f"irreps must be mul_0 x0e + mul_1 x1e + ...: {irreps}")
        ebm = bool(k.get('ebm', False))
        ete, qte = bool(k.get('edge_time_encoding', False)), bool(k.get('query_time_encoding', True))
        if (ebm and (ete or qte)) or (not ebm and not ete and not qte):
            raise NotImplementedError("accelerated path: score head with edge_time_encoding and / or query_time_encoding (the reference "
                                      "constructor's default is query_time_encoding alone), or EBM critic head with both False")
        if tf.get('n_layers', 1) != 1:
            raise NotImplementedError("n_layers != 1")
        if tf.get('cutoff_method', 'edge_attn') != 'edge_attn':
            raise NotImplementedError("cutoff_method != 'edge_attn'")
        if tf.get('use_dst_point_attn', False):
            raise NotImplementedError                                                   # gnn_block.py:196-197
        sh = parse_irreps(tf['irreps_sh'])
        assert all(m == 1 for m, _ in sh) and [l for _, l in sh] == list(range(len(sh)))
        fc = list(tf['fc_neurons'])
        temb = list(k['time_emb_mlp'])
        if fc[0] == -1:
            fc[0] = tf['length_emb_dim'] + (temb[-1] if ete else 0)
        assert fc[0] == tf['length_emb_dim'] + (temb[-1] if ete else 0)
        if fc[1:] not in ([128, 64], [32, 32]):
            raise NotImplementedError(f"fc_neurons {fc}: radial MLPs are instantiated for [*, 128, 64] (panda_* configs, sapien pick_*) "
                                      "and [*, 32, 32] (sapien place_*)")
        if fc[1:] == [32, 32] and fc[0] != (64 if ebm else 128):
            raise NotImplementedError(f"fc_neurons {fc}: the 32-wide radial MLP is instantiated for the score head with a 64-channel "
                                      "time embedding (the sapien place_* configs) and for context-free fields (KeypointExtractor)")
        if temb not in ([256, 128, 64], [512, 256, 128]):
            raise NotImplementedError(f"time_emb_mlp {temb}")
        radii = [None if r is None else float(r) for r in tf['r_cluster_multiscale']]
        if 'n_scales' in tf and tf['n_scales'] is not None:
            assert tf['n_scales'] == len(radii)
        seen_inf = False
        for r in radii:
            if r is None:
                seen_inf = True
            elif seen_inf:
                raise ValueError(f"Finite cluster radius cannot come after infinite cluster radius, {radii}")
        rmin = tf.get('r_mincut_nonscalar_sh', None)
        if rmin is None:
            rmin = 0.01 * radii[0]
        lmr = tf.get('length_enc_max_r', None)
        if radii[-1] is None:
            assert lmr is not None
        elif lmr is not None:
            raise AssertionError("You don't need to provide length_enc_max_r")      # multiscale_tensor_field.py:100
        return cls(irreps=irreps, lmax_sh=len(sh) - 1, num_heads=int(tf['num_heads']), fc_neurons=fc,
                   length_emb_dim=int(tf['length_emb_dim']), radii=radii, r_mincut_nonscalar_sh=float(rmin),
                   length_enc_max_r=float(lmr) if lmr is not None else 0.0, time_emb_mlp=temb,
                   max_time=float(k['max_time']), time_enc_n=float(k.get('time_enc_n', 10000.)),
                   lin_mult=float(k['lin_mult']), ang_mult=float(k['ang_mult']),
                   irreps_mlp_mid=int(tf.get('irreps_mlp_mid', 3)), ebm=ebm,
                   use_src_point_attn=bool(tf.get('use_src_point_attn', False)), query_time_encoding=qte)


# --------------------------------------------------------------------------------------------------
# tensor-product path tables (shared by the packer and the table generator)
# --------------------------------------------------------------------------------------------------

def dtp_paths(irreps: Irreps, ls2: List[int], muls2: List[int], l_out_allowed: List[int]):
    """Creation-order instruction list of DepthwiseTensorProduct (reference
    tensor_product_rescale.py:365-371): [(l1, l2, l3, mul1, mul2)]."""
    paths = []
    for m1, l1 in irreps:
        for m2, l2 in zip(muls2, ls2):
            for l3 in range(abs(l1 - l2), l1 + l2 + 1):
                if l3 in l_out_allowed or l3 == 0:
                    paths.append((l1, l2, l3, m1, m2))
    return paths


def dtp_sorted_out(paths) -> Dict[int, List[int]]:
    """path indices grouped by l3 in sorted-output order (stable by creation index)."""
    out: Dict[int, List[int]] = {}
    for p, (l1, l2, l3, m1, m2) in enumerate(paths):
        out.setdefault(l3, []).append(p)
    return out


# --------------------------------------------------------------------------------------------------
# schema
# --------------------------------------------------------------------------------------------------

def param_spec(cfg: HeadConfig) -> List[Tuple[str, Tuple[int, ...], str, float]]:
    """[(name, shape, kind, scale)], kind in {'linear_w','linear_b','tp_w','zeros','ones','const','xavier'}.
    `scale` is fan_in for linear/tp kinds, the constant for 'const'."""
    S: List[Tuple[str, Tuple[int, ...], str, float]] = []
    muls, L = cfg.muls, cfg.lmax
    D = cfg.dim
    n0 = muls[0]
    te = cfg.time_emb_mlp
    for n in range(cfg.n_scales):
        li = 0
        for i in range(1, len(te)):
            S.append((f"time_mlps_multiscale.{n}.{li}.weight", (te[i], te[i - 1]), 'linear_w', te[i - 1]))
            S.append((f"time_mlps_multiscale.{n}.{li}.bias", (te[i],), 'linear_b', te[i - 1]))
            li += 2 if i != len(te) - 1 else 1
    qte = cfg.query_time_encoding
    if qte:                              # score_head.py:64-70: same shape as one time MLP
        li = 0
        for i in range(1, len(te)):
            S.append((f"query_time_mlp.{li}.weight", (te[i], te[i - 1]), 'linear_w', te[i - 1]))
            S.append((f"query_time_mlp.{li}.bias", (te[i],), 'linear_b', te[i - 1]))
            li += 2 if i != len(te) - 1 else 1
    ktf = "key_tensor_field"
    F0 = cfg.fc_neurons[0]
    dimL = cfg.length_emb_dim
    for n, r in enumerate(cfg.radii):
        if r is not None:
            pm = f"{ktf}.graph_parsers.{n}.length_enc.param_module"
            S.append((f"{pm}.std_logit", (1, dimL), 'const', math.log(math.exp(2.0 / dimL) - 1)))
            S.append((f"{pm}.weight_logit", (1, dimL), 'const', -math.log(4.0 / 1. - 1)))
            S.append((f"{pm}.mean", (1, dimL), 'linspace', 0.0))
        S.append((f"{ktf}.edge_scalars_pre_linears.{n}.0.weight", (F0, F0), 'linear_w', F0))
        S.append((f"{ktf}.edge_scalars_pre_linears.{n}.0.bias", (F0,), 'linear_b', F0))
    blk = f"{ktf}.gnn_block_init"
    nirr = sum(muls)
    S.append((f"{blk}.prenorm_src.affine_weight", (nirr,), 'ones', 0))
    S.append((f"{blk}.prenorm_src.affine_bias", (n0,), 'zeros', 0))
    S.append((f"{blk}.linear_src.tp.weight", (sum(m * m for m in muls),), 'tp_w:' + ','.join(f"{m*m}:{m}" for m in muls), 0))
    if not qte:
        S.append((f"{blk}.linear_src.bias.0", (n0,), 'zeros', 0))
    else:                                # use_dst_feature=True (gnn_block.py:109-130): the destination features are the te[-1] time scalars
        tq = te[-1]
        S.append((f"{blk}.skip_1.skip.tp.weight", (tq * n0,), f'tp_w:{tq * n0}:{tq}', 0))
        S.append((f"{blk}.skip_1.skip.bias.0", (n0,), 'zeros', 0))
        S.append((f"{blk}.prenorm_dst.affine_weight", (tq,), 'ones', 0))
        S.append((f"{blk}.prenorm_dst.affine_bias", (tq,), 'zeros', 0))
        S.append((f"{blk}.linear_dst.tp.weight", (tq * n0,), f'tp_w:{tq * n0}:{tq}', 0))
        S.append((f"{blk}.linear_dst.bias.0", (n0,), 'zeros', 0))
    ga = f"{blk}.ga"
    # radial profile
    sh_ls = list(range(cfg.lmax_sh + 1))
    paths = dtp_paths(cfg.irreps, sh_ls, [1] * len(sh_ls), list(range(L + 1)))
    wn = sum(p[3] for p in paths)
    ch = cfg.fc_neurons + [wn]
    idx = 0
    for i in range(1, len(ch)):
        last = i == len(ch) - 1
        S.append((f"{ga}.sep_act.dtp_rad.net.{idx}.weight", (ch[i], ch[i - 1]), 'linear_w', ch[i - 1]))
        if not last:
            S.append((f"{ga}.sep_act.dtp_rad.net.{idx}.bias", (ch[i],), 'linear_b', ch[i - 1]))
            S.append((f"{ga}.sep_act.dtp_rad.net.{idx + 1}.weight", (ch[i],), 'ones', 0))
            S.append((f"{ga}.sep_act.dtp_rad.net.{idx + 1}.bias", (ch[i],), 'zeros', 0))
            idx += 3
    S.append((f"{ga}.sep_act.dtp_rad.offset", (wn,), 'linear_b', ch[-2]))
    by_l = dtp_sorted_out(paths)
    mul_dtp = [sum(paths[p][3] for p in by_l.get(l, [])) for l in range(L + 1)]     # 112,192,176
    gates = sum(muls[1:])
    lin_out = [n0 + gates] + muls[1:]
    S.append((f"{ga}.sep_act.lin.tp.weight", (sum(a * b for a, b in zip(mul_dtp, lin_out)),),
              'tp_w:' + ','.join(f"{a*b}:{a}" for a, b in zip(mul_dtp, lin_out)), 0))
    S.append((f"{ga}.sep_act.lin.bias.0", (lin_out[0],), 'zeros', 0))
    a_blocks = [paths[p][3] for p in by_l[0]]
    S.append((f"{ga}.sep_alpha.tp.weight", (sum(a * n0 for a in a_blocks),),
              'tp_w:' + ','.join(f"{a*n0}:{sum(a_blocks)}" for a in a_blocks), 0))
    S.append((f"{ga}.sep_alpha.bias.0", (n0,), 'zeros', 0))
    S.append((f"{ga}.sep_value.dtp.tp.weight", (wn,), 'tp_w:' + f"{wn}:1", 0))
    S.append((f"{ga}.sep_value.lin.tp.weight", (sum(a * b for a, b in zip(mul_dtp, muls)),),
              'tp_w:' + ','.join(f"{a*b}:{a}" for a, b in zip(mul_dtp, muls)), 0))
    S.append((f"{ga}.sep_value.lin.bias.0", (n0,), 'zeros', 0))
    S.append((f"{ga}.alpha_dot", (1, cfg.num_heads, n0 // cfg.num_heads), 'xavier', 0))
    S.append((f"{ga}.proj.tp.weight", (sum(m * m for m in muls),), 'tp_w:' + ','.join(f"{m*m}:{m}" for m in muls), 0))
    S.append((f"{ga}.proj.bias.0", (n0,), 'zeros', 0))
    S.append((f"{blk}.post_norm.affine_weight", (nirr,), 'ones', 0))
    S.append((f"{blk}.post_norm.affine_bias", (n0,), 'zeros', 0))
    K = cfg.irreps_mlp_mid
    mid = [m * K for m in muls]
    f1_out = [mid[0] + sum(mid[1:])] + mid[1:]
    S.append((f"{blk}.ffn.fctp_1.tp.weight", (sum(a * b for a, b in zip(muls, f1_out)),),
              'tp_w:' + ','.join(f"{a*b}:{a}" for a, b in zip(muls, f1_out)), 0))
    S.append((f"{blk}.ffn.fctp_1.bias.0", (f1_out[0],), 'zeros', 0))
    S.append((f"{blk}.ffn.fctp_2.tp.weight", (sum(a * b for a, b in zip(mid, muls)),),
              'tp_w:' + ','.join(f"{a*b}:{a}" for a, b in zip(mid, muls)), 0))
    S.append((f"{blk}.ffn.fctp_2.bias.0", (n0,), 'zeros', 0))
    # score tensor products: in1 = in2 = irreps, l_out in {0, 1}
    spaths = dtp_paths(cfg.irreps, [l for _, l in cfg.irreps], muls, [0, 1])
    sby = dtp_sorted_out(spaths)
    n_pre = muls[1]
    smul = [sum(spaths[p][3] for p in sby.get(l, [])) for l in (0, 1)]
    for name in (() if cfg.ebm else ("lin_vel_tp", "ang_vel_tp")):
        S.append((f"{name}.dtp.tp.weight", (sum(p[3] * p[4] for p in spaths),),
                  'tp_w:' + ','.join(f"{p[3]*p[4]}:{p[4]}" for p in spaths), 0))
        lo = [1 + n_pre, n_pre]
        S.append((f"{name}.lin.tp.weight", (sum(a * b for a, b in zip(smul, lo)),),
                  'tp_w:' + ','.join(f"{a*b}:{a}" for a, b in zip(smul, lo)), 0))
        S.append((f"{name}.lin.bias.0", (lo[0],), 'zeros', 0))
    return S


def init_params(cfg: HeadConfig, seed: int = 2, randomize_all: bool = False,
                dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """Seeded random-init weights of the reference architecture (no checkpoints are available: every *.pt
    in the reference is an LFS stub).  nn.Linear: U(+-1/sqrt(fan_in)); e3nn TP weights: N(0,1)/sqrt(fan_in)
    (reference tensor_product_rescale.py:94-120); biases / LN affine at their reference init unless
    `randomize_all` (parity tests use that so that no term is trivially zero)."""
    g = torch.Generator().manual_seed(seed)
    P: Dict[str, torch.Tensor] = {}
    for name, shape, kind, scale in param_spec(cfg):
        P[name] = _init_one(g, shape, kind, scale, randomize_all).to(dtype)
    return P


def _init_one(g, shape, kind, scale, randomize_all):
    if True:
        if kind in ('linear_w', 'linear_b'):
            b = 1.0 / math.sqrt(scale)
            t = (torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1) * b
        elif kind.startswith('tp_w:'):
            parts = []
            for blk in kind[5:].split(','):
                n, fan = blk.split(':')
                parts.append(torch.randn(int(n), generator=g, dtype=torch.float64) / math.sqrt(float(fan)))
            t = torch.cat(parts)
            assert t.numel() == shape[0], (name, t.numel(), shape)
        elif kind == 'zeros':
            t = torch.zeros(shape, dtype=torch.float64)
            if randomize_all:
                t = torch.randn(shape, generator=g, dtype=torch.float64) * 0.1
        elif kind == 'ones':
            t = torch.ones(shape, dtype=torch.float64)
            if randomize_all:
                t = t + torch.randn(shape, generator=g, dtype=torch.float64) * 0.1
        elif kind == 'const':
            t = torch.full(shape, scale, dtype=torch.float64)
            if randomize_all:
                t = t + torch.randn(shape, generator=g, dtype=torch.float64) * 0.05
        elif kind == 'linspace':
            t = torch.linspace(0.0, 1.0, shape[-1] + 2, dtype=torch.float64)[1:-1].reshape(shape)
        elif kind == 'xavier':
            fan_in, fan_out = shape[1] * shape[2], shape[0] * shape[2]
            b = math.sqrt(6.0 / (fan_in + fan_out))
            t = (torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1) * b
        else:
            raise ValueError(kind)
        return t


def n_params(cfg: HeadConfig) -> int:
    return sum(int(np.prod(s)) for _, s, _, _ in param_spec(cfg))


# --------------------------------------------------------------------------------------------------
# one layer of the UNet feature extractor (the whole extractor: unet.py)
# --------------------------------------------------------------------------------------------------

def unet_layer_param_spec(irreps: Irreps, fc_neurons: List[int], num_heads: int = 4, lmax_sh: Optional[int] = None,
                          irreps_mlp_mid: int = 3, irreps_src: Optional[Irreps] = None,
                          mid_muls: Optional[List[int]] = None) -> List[Tuple[str, Tuple[int, ...], str, float]]:
    """Schema of the ModuleDict {'radial': GaussianRadialBasisLayerFiniteCutoff, 'gnn': block.EquiformerBlock} the reference builds
    for every UNet layer (unet_feature_extractor.py:141-156; names = its state_dict keys below e.g. ``down_blocks.3.pool_layer.``).
    ``irreps`` = irreps_dst = the block's irreps_emb; ``irreps_src`` (default: the same) only shapes norm_1_src / linear_src
    (block.py:108-109).  Same tuple format as ``param_spec``."""
    S: List[Tuple[str, Tuple[int, ...], str, float]] = []
    muls = [m for m, _ in irreps]
    muls_src = muls if irreps_src is None else [m for m, _ in irreps_src]
    L, n0, nirr = len(muls) - 1, muls[0], sum(muls)
    lmax_sh = L if lmax_sh is None else lmax_sh
    nb = fc_neurons[0]
    S.append(("radial.mean", (1, nb), 'linspace', 0.0))                                             # radial_func.py:242-243
    S.append(("radial.std_logit", (1, nb), 'const', math.log(math.exp(2.0 / nb) - 1)))              # :248-249
    S.append(("radial.weight_logit", (1, nb), 'const', -math.log(4.0 / 1. - 1)))                    # :251-252
    g = "gnn"
    sq = 'tp_w:' + ','.join(f"{m*m}:{m}" for m in muls)
    S.append((f"{g}.norm_1_src.affine_weight", (sum(muls_src),), 'ones', 0)); S.append((f"{g}.norm_1_src.affine_bias", (muls_src[0],), 'zeros', 0))
    S.append((f"{g}.linear_src.tp.weight", (sum(a * b for a, b in zip(muls_src, muls)),),
              'tp_w:' + ','.join(f"{a*b}:{a}" for a, b in zip(muls_src, muls)), 0))                    # src_bias=False
    S.append((f"{g}.norm_1_dst.affine_weight", (nirr,), 'ones', 0)); S.append((f"{g}.norm_1_dst.affine_bias", (n0,), 'zeros', 0))
    S.append((f"{g}.linear_dst.tp.weight", (sum(m * m for m in muls),), sq, 0))
    S.append((f"{g}.linear_dst.bias.0", (n0,), 'zeros', 0))
    ga = f"{g}.ga"
    sh_ls = list(range(lmax_sh + 1))
    paths = dtp_paths(irreps, sh_ls, [1] * len(sh_ls), list(range(L + 1)))
    wn = sum(p[3] for p in paths)
    ch = list(fc_neurons) + [wn]
    idx = 0
    for i in range(1, len(ch)):
        last = i == len(ch) - 1
        S.append((f"{ga}.sep_act.dtp_rad.net.{idx}.weight", (ch[i], ch[i - 1]), 'linear_w', ch[i - 1]))
        if not last:
            S.append((f"{ga}.sep_act.dtp_rad.net.{idx}.bias", (ch[i],), 'linear_b', ch[i - 1]))
            S.append((f"{ga}.sep_act.dtp_rad.net.{idx + 1}.weight", (ch[i],), 'ones', 0))
            S.append((f"{ga}.sep_act.dtp_rad.net.{idx + 1}.bias", (ch[i],), 'zeros', 0))
            idx += 3
    S.append((f"{ga}.sep_act.dtp_rad.offset", (wn,), 'linear_b', ch[-2]))
    by_l = dtp_sorted_out(paths)
    mul_dtp = [sum(paths[p][3] for p in by_l.get(l, [])) for l in range(L + 1)]
    lin_out = [n0 + sum(muls[1:])] + muls[1:]
    S.append((f"{ga}.sep_act.lin.tp.weight", (sum(a * b for a, b in zip(mul_dtp, lin_out)),),
              'tp_w:' + ','.join(f"{a*b}:{a}" for a, b in zip(mul_dtp, lin_out)), 0))
    S.append((f"{ga}.sep_act.lin.bias.0", (lin_out[0],), 'zeros', 0))
    a_blocks = [paths[p][3] for p in by_l[0]]
    S.append((f"{ga}.sep_alpha.tp.weight", (sum(a * n0 for a in a_blocks),), 'tp_w:' + ','.join(f"{a*n0}:{sum(a_blocks)}" for a in a_blocks), 0))
    S.append((f"{ga}.sep_alpha.bias.0", (n0,), 'zeros', 0))
    S.append((f"{ga}.sep_value.dtp.tp.weight", (wn,), 'tp_w:' + f"{wn}:1", 0))
    S.append((f"{ga}.sep_value.lin.tp.weight", (sum(a * b for a, b in zip(mul_dtp, muls)),),
              'tp_w:' + ','.join(f"{a*b}:{a}" for a, b in zip(mul_dtp, muls)), 0))
    S.append((f"{ga}.sep_value.lin.bias.0", (n0,), 'zeros', 0))
    S.append((f"{ga}.alpha_dot", (1, num_heads, n0 // num_heads), 'xavier', 0))
    S.append((f"{ga}.proj.tp.weight", (sum(m * m for m in muls),), sq, 0))
    S.append((f"{ga}.proj.bias.0", (n0,), 'zeros', 0))
    S.append((f"{g}.norm_2.affine_weight", (nirr,), 'ones', 0)); S.append((f"{g}.norm_2.affine_bias", (n0,), 'zeros', 0))
    mid = [m * irreps_mlp_mid for m in muls] if mid_muls is None else list(mid_muls)       # (mid_muls: the kernels' padded hidden widths, unet_pad.WIDE_HID)
    f1_out = [mid[0] + sum(mid[1:])] + mid[1:]
    S.append((f"{g}.ffn.fctp_1.tp.weight", (sum(a * b for a, b in zip(muls, f1_out)),), 'tp_w:' + ','.join(f"{a*b}:{a}" for a, b in zip(muls, f1_out)), 0))
    S.append((f"{g}.ffn.fctp_1.bias.0", (f1_out[0],), 'zeros', 0))
    S.append((f"{g}.ffn.fctp_2.tp.weight", (sum(a * b for a, b in zip(mid, muls)),), 'tp_w:' + ','.join(f"{a*b}:{a}" for a, b in zip(mid, muls)), 0))
    S.append((f"{g}.ffn.fctp_2.bias.0", (n0,), 'zeros', 0))
    return S


def init_from_spec(spec, seed: int = 2, randomize_all: bool = False, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """seeded init of any schema in the ``param_spec`` tuple format (see ``init_params``)"""
    class _C:            # init_params only needs `param_spec(cfg)`: feed it the ready-made list
        pass
    g = torch.Generator().manual_seed(seed)
    P: Dict[str, torch.Tensor] = {}
    for name, shape, kind, scale in spec:
        P[name] = _init_one(g, shape, kind, scale, randomize_all).to(dtype)
    return P
