"""Zero-padded embedding of the narrow UNet layers into the instantiated kernel shape.

The fused kernels are built for ``64x0e+32x1e+16x2e`` with a ``[64, 32, 32]`` radial network.  The two fine levels of every shipped
UNet use ``32x0e+16x1e+8x2e`` with ``[32, 16, 16]`` (``configs/*/*/score_model_configs.yaml: irreps_emb[:2]``).  Every operation of a
layer is linear per channel, a gate / activation that maps 0 to 0, a LayerNorm, or the per-head softmax — so a narrow layer is
EXACTLY a wide layer whose extra channels carry zeros, provided that

* the true channels are placed so that the kernel's head assignment (head = channel // (mul / 4)) equals the reference's
  (head = channel // (true_mul / 4)) AND every group of four wide channels holds the same number of true ones (the lmax-3 and the narrow
  instantiations of the kernels skip the lane-local work on the others): channel c goes to ``head * (M/4) + 4 * (k // q) + k % q`` with
  ``k = c % (m/4)``, ``q = 4 m / M`` (``place``; half-filled blocks 0, 1, 4, 5, ..., quarter-filled 0, 4, 8, ...);
* every weight touching a padded channel is zero (this module builds those parameter tensors, same names as the wide schema);
* the LayerNorms take their statistics over the true channels only (told to the kernels: ``dedf_config.unet_valid / unet_fc_valid``);
* the radial basis is normalised by sqrt(true num_basis).

Nothing here runs per edge: it turns a reference-shaped state dict into the wide one once, and pads / un-pads feature tensors at the
layer boundary.
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import torch

from .params import dtp_paths, dtp_sorted_out

WIDE = [64, 32, 16]          # instantiated multiplicities (lmax 2)
WIDE3 = [64, 32, 16, 16]     # lmax 3: the reference's widest 3e block, 8x3e, itself runs zero-padded to one 16-channel chunk (csrc/dedf_net.h::mul_of)
WIDE_HID = [192, 96, 48, 32]   # FFN hidden multiplicities of the kernels (irreps_mlp_mid = 3; 24x3e padded to one 32-row tile: dedf_net.h::hid_of)
WIDE_FC = [64, 32, 32]
NARROW3 = [32, 16, 8, 4]       # the narrow level shape of the panda UNets (levels 0-1), per degree: dedf_config.unet_narrow
HEADS = 4


def wide_of(muls) -> List[int]:
    """kernel multiplicities for true multiplicities `muls` (lmax = len(muls) - 1)"""
    return WIDE3[:len(muls)] if len(muls) == 4 else WIDE[:len(muls)]


def wide_dim(L: int) -> int:
    return sum(M * (2 * l + 1) for l, M in enumerate(WIDE3[:L + 1]))


def place(m: int, M: int) -> torch.Tensor:
    """index map true channel -> wide channel of one irreps block (identity when m == M).  Head h's m / 4 channels stay inside the head's M / 4 wide
    channels (the kernels' head of a channel is channel // (M / 4)); inside a head they go to the positions p with p % 4 < q, q = 4 m / M
    (half-filled blocks: 0, 1, 4, 5, 8, 9, ...; quarter-filled: 0, 4, 8, ...), so that every group of four wide channels holds the same number of
    true ones: the narrow / lmax-3 instantiations of the kernels skip the lane-local work on the others (csrc/dedf_net.h::pad_live)."""
    c = torch.arange(m)
    if m == M:
        return c
    q = 4 * m // M
    if not (m % HEADS == 0 and M % (4 * HEADS) == 0 and q in (1, 2) and 4 * m == q * M):
        raise ValueError(f"unet_pad.place: a block of {m} true channels cannot be embedded in {M} kernel channels; supported: m == M, m == M / 2, m == M / 4 "
                         f"(with m a multiple of {HEADS} and M of {4 * HEADS})")
    k = c % (m // HEADS)
    return (c // (m // HEADS)) * (M // HEADS) + 4 * (k // q) + k % q


_PLACE_DEV = {}


def place_on(m: int, M: int, device) -> torch.Tensor:
    """`place` as a device tensor, built once per (m, M, device)"""
    key = (m, M, str(device))
    if key not in _PLACE_DEV:
        _PLACE_DEV[key] = place(m, M).to(device)
    return _PLACE_DEV[key]


def _first(m: int) -> torch.Tensor:
    return torch.arange(m)


def pad_features(f: torch.Tensor, muls: Sequence[int]) -> torch.Tensor:
    """(N, sum m_l (2l+1)) in the true irreps -> (N, 240 | 352) in the wide layout ([mul][m] per block, true channels at `place`)"""
    W = wide_of(muls)
    if list(muls) == W:
        return f
    out = f.new_zeros(f.shape[0], sum(M * (2 * l + 1) for l, M in enumerate(W)))
    o_t = o_w = 0
    for l, (m, M) in enumerate(zip(muls, W)):
        d = 2 * l + 1
        idx = place_on(m, M, f.device)
        out[:, o_w:o_w + M * d].view(-1, M, d)[:, idx, :] = f[:, o_t:o_t + m * d].reshape(-1, m, d)
        o_t += m * d
        o_w += M * d
    return out


def unpad_features(f: torch.Tensor, muls: Sequence[int]) -> torch.Tensor:
    W = wide_of(muls)
    if list(muls) == W:
        return f
    parts, o_w = [], 0
    for l, (m, M) in enumerate(zip(muls, W)):
        d = 2 * l + 1
        parts.append(f[:, o_w:o_w + M * d].reshape(-1, M, d)[:, place_on(m, M, f.device), :].reshape(-1, m * d))
        o_w += M * d
    return torch.cat(parts, dim=-1)


def _blocks(flat: torch.Tensor, shapes):
    out, o = [], 0
    for a, b in shapes:
        out.append(flat[o:o + a * b].reshape(a, b))
        o += a * b
    assert o == flat.numel(), (o, flat.numel())
    return out


def _embed(block: torch.Tensor, rows: torch.Tensor, cols: torch.Tensor, R: int, Cn: int) -> torch.Tensor:
    out = block.new_zeros(R, Cn)
    out[rows[:, None], cols[None, :]] = block
    return out


def _vec(v: torch.Tensor, idx: torch.Tensor, n: int, fill: float = 0.0) -> torch.Tensor:
    out = v.new_full((n,), fill)
    out[idx] = v.reshape(-1)
    return out


def expand_layer_params(P: Dict[str, torch.Tensor], muls: Sequence[int], fc: Sequence[int], muls_src: Sequence[int]) -> Dict[str, torch.Tensor]:
    """reference-shaped parameters of a {radial, gnn} layer with true multiplicities `muls` (dst / emb), `muls_src` and radial widths `fc`
    -> the equivalent parameters of the wide schema (``params.unet_layer_param_spec([(64,0),(32,1),(16,2)], [64,32,32])``; at lmax 3 the wide
    schema is 64x0e+32x1e+16x2e+16x3e with FFN hidden multiplicities 192 / 96 / 48 / 32, the library's ``dedf_param_*`` list)"""
    m, ms, M = list(muls), list(muls_src), wide_of(list(muls))
    HID = WIDE_HID[:len(m)]
    nb, h1, h2 = fc
    NB, H1, H2 = WIDE_FC
    L = len(m) - 1
    pl = [place(a, b) for a, b in zip(m, M)]                  # emb-type features of the block
    pls = [place(a, b) for a, b in zip(ms, M)]                # source features
    Q: Dict[str, torch.Tensor] = {}
    f32 = lambda t: t.detach().to(torch.float32).cpu()
    P = {k: f32(v) for k, v in P.items()}
    # ---- radial basis: padded basis functions get weight sigmoid(-1e4) = 0
    Q["radial.mean"] = _vec(P["radial.mean"], _first(nb), NB, 0.5).reshape(1, NB)
    Q["radial.std_logit"] = _vec(P["radial.std_logit"], _first(nb), NB, 0.0).reshape(1, NB)
    Q["radial.weight_logit"] = _vec(P["radial.weight_logit"], _first(nb), NB, -1.0e4).reshape(1, NB)
    g = "gnn"
    irr_w = [(M[l], l) for l in range(L + 1)]
    sh_ls = list(range(L + 1))
    paths_t = dtp_paths([(m[l], l) for l in range(L + 1)], sh_ls, [1] * (L + 1), list(range(L + 1)))
    paths_w = dtp_paths(irr_w, sh_ls, [1] * (L + 1), list(range(L + 1)))
    # dead norms (block.py:149-153): any values
    Q[f"{g}.norm_1_src.affine_weight"] = torch.ones(sum(M)); Q[f"{g}.norm_1_src.affine_bias"] = torch.zeros(M[0])
    Q[f"{g}.norm_1_dst.affine_weight"] = torch.ones(sum(M)); Q[f"{g}.norm_1_dst.affine_bias"] = torch.zeros(M[0])
    # per-degree square / rectangular maps [in][out]
    def per_l(name, rows_t, rows_pl, cols_pl=pl):
        bl = _blocks(P[name], [(rows_t[l], m[l]) for l in range(L + 1)])
        Q[name] = torch.cat([_embed(bl[l], rows_pl[l], cols_pl[l], M[l], M[l]).reshape(-1) for l in range(L + 1)])
    per_l(f"{g}.linear_src.tp.weight", ms, pls)
    per_l(f"{g}.linear_dst.tp.weight", m, pl)
    Q[f"{g}.linear_dst.bias.0"] = _vec(P[f"{g}.linear_dst.bias.0"], pl[0], M[0])
    ga, rad = f"{g}.ga", f"{g}.ga.sep_act.dtp_rad"
    # ---- radial MLP: first-part placement of the hidden channels; last layer rows = DTP channels at `place`
    Q[f"{rad}.net.0.weight"] = _embed(P[f"{rad}.net.0.weight"], _first(h1), _first(nb), H1, NB)
    Q[f"{rad}.net.0.bias"] = _vec(P[f"{rad}.net.0.bias"], _first(h1), H1)
    Q[f"{rad}.net.1.weight"] = _vec(P[f"{rad}.net.1.weight"], _first(h1), H1)
    Q[f"{rad}.net.1.bias"] = _vec(P[f"{rad}.net.1.bias"], _first(h1), H1)
    Q[f"{rad}.net.3.weight"] = _embed(P[f"{rad}.net.3.weight"], _first(h2), _first(h1), H2, H1)
    Q[f"{rad}.net.3.bias"] = _vec(P[f"{rad}.net.3.bias"], _first(h2), H2)
    Q[f"{rad}.net.4.weight"] = _vec(P[f"{rad}.net.4.weight"], _first(h2), H2)
    Q[f"{rad}.net.4.bias"] = _vec(P[f"{rad}.net.4.bias"], _first(h2), H2)
    wn_w = sum(p[3] for p in paths_w)
    # flat DTP weight index (creation order): true -> wide
    wmap = []
    st, sw = 0, 0
    for pt, pw in zip(paths_t, paths_w):
        wmap.append(sw + pl[pt[0]])
        st += pt[3]; sw += pw[3]
    wmap = torch.cat(wmap)
    Q[f"{rad}.net.6.weight"] = _embed(P[f"{rad}.net.6.weight"], wmap, _first(h2), wn_w, H2)
    Q[f"{rad}.offset"] = _vec(P[f"{rad}.offset"], wmap, wn_w)
    Q[f"{ga}.sep_value.dtp.tp.weight"] = _vec(P[f"{ga}.sep_value.dtp.tp.weight"], wmap, wn_w)
    # ---- sorted DTP channels per output degree (K index of sep_act.lin / sep_value.lin): true -> wide
    by_t, by_w = dtp_sorted_out(paths_t), dtp_sorted_out(paths_w)
    kmap, k_t, k_w = [], [], []
    for l3 in range(L + 1):
        idx, ow = [], 0
        for p in by_t[l3]:
            idx.append(ow + pl[paths_t[p][0]])
            ow += paths_w[p][3]
        kmap.append(torch.cat(idx)); k_t.append(sum(paths_t[p][3] for p in by_t[l3])); k_w.append(ow)
    # lin0 rows: [scalars | gates of l = 1 | gates of l = 2 ...]
    def lin0_map(mm, MM, pls_):
        idx, ot, ow = [pls_[0]], mm[0], MM[0]
        for l in range(1, L + 1):
            idx.append(ow + pls_[l]); ow += MM[l]
        return torch.cat(idx), sum(mm), sum(MM)
    o0, n0_t, n0_w = lin0_map(m, M, pl)
    lin_t = [(k_t[0], n0_t)] + [(k_t[l], m[l]) for l in range(1, L + 1)]
    bl = _blocks(P[f"{ga}.sep_act.lin.tp.weight"], lin_t)
    Q[f"{ga}.sep_act.lin.tp.weight"] = torch.cat([_embed(bl[0], kmap[0], o0, k_w[0], n0_w).reshape(-1)] +
                                                 [_embed(bl[l], kmap[l], pl[l], k_w[l], M[l]).reshape(-1) for l in range(1, L + 1)])
    Q[f"{ga}.sep_act.lin.bias.0"] = _vec(P[f"{ga}.sep_act.lin.bias.0"], o0, n0_w)
    # sep_alpha: one block per scalar path (un-simplified input), alpha channel a = h * (m0/4) + k -> place(m0, M0)[a]
    a_t = [(paths_t[p][3], m[0]) for p in by_t[0]]
    bl = _blocks(P[f"{ga}.sep_alpha.tp.weight"], a_t)
    Q[f"{ga}.sep_alpha.tp.weight"] = torch.cat([_embed(b_, pl[paths_t[p][0]], pl[0], paths_w[p][3], M[0]).reshape(-1) for b_, p in zip(bl, by_t[0])])
    Q[f"{ga}.sep_alpha.bias.0"] = _vec(P[f"{ga}.sep_alpha.bias.0"], pl[0], M[0])
    ad = P[f"{ga}.alpha_dot"].reshape(HEADS, m[0] // HEADS)
    adw = torch.zeros(HEADS, M[0] // HEADS); adw[:, pl[0][: m[0] // HEADS]] = ad          # (head 0's positions inside a head: `place`)
    Q[f"{ga}.alpha_dot"] = adw.reshape(1, HEADS, M[0] // HEADS)
    bl = _blocks(P[f"{ga}.sep_value.lin.tp.weight"], [(k_t[l], m[l]) for l in range(L + 1)])
    Q[f"{ga}.sep_value.lin.tp.weight"] = torch.cat([_embed(bl[l], kmap[l], pl[l], k_w[l], M[l]).reshape(-1) for l in range(L + 1)])
    Q[f"{ga}.sep_value.lin.bias.0"] = _vec(P[f"{ga}.sep_value.lin.bias.0"], pl[0], M[0])
    per_l(f"{ga}.proj.tp.weight", m, pl)
    Q[f"{ga}.proj.bias.0"] = _vec(P[f"{ga}.proj.bias.0"], pl[0], M[0])
    # ---- norm_2 (masked statistics in the kernel; padded affine weights 0)
    wt, ww, ot, ow = P[f"{g}.norm_2.affine_weight"], torch.zeros(sum(M)), 0, 0
    for l in range(L + 1):
        ww[ow + pl[l]] = wt[ot:ot + m[l]]
        ot += m[l]; ow += M[l]
    Q[f"{g}.norm_2.affine_weight"] = ww
    Q[f"{g}.norm_2.affine_bias"] = _vec(P[f"{g}.norm_2.affine_bias"], pl[0], M[0])
    # ---- FFN: hidden channels first-part; rows of fctp_1's 0e block: [3 m0 scalars | 3 m1 gates | 3 m2 gates]
    K3 = 3
    def f1_map(mm, HH):
        idx, ow = [_first(K3 * mm[0])], HH[0]
        for l in range(1, L + 1):
            idx.append(ow + _first(K3 * mm[l])); ow += HH[l]
        return torch.cat(idx), ow
    f1o, f1n_w = f1_map(m, HID)
    f1n_t = K3 * sum(m)
    bl = _blocks(P[f"{g}.ffn.fctp_1.tp.weight"], [(m[0], f1n_t)] + [(m[l], K3 * m[l]) for l in range(1, L + 1)])
    Q[f"{g}.ffn.fctp_1.tp.weight"] = torch.cat([_embed(bl[0], pl[0], f1o, M[0], f1n_w).reshape(-1)] +
                                                [_embed(bl[l], pl[l], _first(K3 * m[l]), M[l], HID[l]).reshape(-1) for l in range(1, L + 1)])
    Q[f"{g}.ffn.fctp_1.bias.0"] = _vec(P[f"{g}.ffn.fctp_1.bias.0"], f1o, f1n_w)
    bl = _blocks(P[f"{g}.ffn.fctp_2.tp.weight"], [(K3 * m[l], m[l]) for l in range(L + 1)])
    Q[f"{g}.ffn.fctp_2.tp.weight"] = torch.cat([_embed(bl[l], _first(K3 * m[l]), pl[l], HID[l], M[l]).reshape(-1) for l in range(L + 1)])
    Q[f"{g}.ffn.fctp_2.bias.0"] = _vec(P[f"{g}.ffn.fctp_2.bias.0"], pl[0], M[0])
    return Q
