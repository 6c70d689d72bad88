"""Stage-by-stage parity of the HIP path against the CPU oracle (GPU required).  Used by tests/test_gpu_parity.py and
runnable on its own:  python tests/stage_check.py [lmax] [nT] [n_scene] [n_grasp]"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from diffusion_edf_amd import params, synthetic  # noqa: E402
from diffusion_edf_amd.gnn_data import FeaturedPoints  # noqa: E402
from diffusion_edf_amd.score_head import ScoreModelHead  # noqa: E402
from oracle import restatement as R  # noqa: E402


KERNEL_MULS = [64, 32, 16, 16]       # multiplicities the kernels run (dedf_net.h::mul_of): the reference's 8x3e is zero-padded to 16x3e


def pad_pos(l, c, muls):
    """kernel channel of true channel c of degree l (dedf_net.h::pad_pos: per-head placement)"""
    tm, km = muls[l] // 4, KERNEL_MULS[l] // 4
    return (c // tm) * km + c % tm


def int_to_ref_perm(muls):
    """perm such that ref_layout (true multiplicities) = internal[..., perm]; internal = the kernels' [l][m][kernel channel]"""
    perm, off = [], 0
    for l, m in enumerate(muls):
        d, M = 2 * l + 1, KERNEL_MULS[l]
        for c in range(m):
            for k in range(d):
                perm.append(off + k * M + pad_pos(l, c, muls))
        off += M * d
    return torch.tensor(perm, dtype=torch.long)


def kernel_ref_to_true_perm(muls):
    """perm such that true ref layout [l][c][m] = kernel ref layout [l][kernel channel][m][..., perm]  (the `msg` buffer)"""
    perm, off = [], 0
    for l, m in enumerate(muls):
        d, M = 2 * l + 1, KERNEL_MULS[l]
        for c in range(m):
            for k in range(d):
                perm.append(off + pad_pos(l, c, muls) * d + k)
        off += M * d
    return torch.tensor(perm, dtype=torch.long)


def dtp_weight_perm(muls):
    """perm such that the true flat depth-wise-TP weights (creation order, tensor_product_rescale.py:365-371) = kernel-shape weights[..., perm]"""
    L = len(muls) - 1
    perm, off = [], 0
    for l1 in range(L + 1):
        for l2 in range(L + 1):
            for l3 in range(abs(l1 - l2), min(L, l1 + l2) + 1):
                perm += [off + pad_pos(l1, u, muls) for u in range(muls[l1])]
                off += KERNEL_MULS[l1]
    return torch.tensor(perm, dtype=torch.long)


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def build_case(lmax=2, nT=6, n_scene=512, n_grasp=100, seed=0, radii=(5., 10., 20., None), static_kp=False, identity_pose=True,
               near=True, query_time_encoding=False, edge_time_encoding=True):
    kw = synthetic.score_head_kwargs(lmax, radii=radii, query_time_encoding=query_time_encoding, edge_time_encoding=edge_time_encoding)
    cfg = params.HeadConfig.from_kwargs(kw)
    P = params.init_params(cfg, seed=2, randomize_all=True)
    keys = synthetic.make_key_clouds(cfg, n_scene, seed=seed)
    query = synthetic.make_query(cfg, n_grasp, seed=seed, static_keypoints=static_kp)
    Ts = synthetic.make_poses(nT, seed=1, near_object=near)
    if identity_pose and nT > 1:
        Ts[0] = torch.tensor([1., 0, 0, 0, 0., 0., 12.], dtype=torch.float64)     # notebook seed quaternion (YXY quirk)
    # (float32-representable: the C ABI takes the diffusion time as float32 -- the reference's `time` tensor -- and the sinusoidal encoding multiplies
    #  it by time_enc_n = 10 000: an oracle evaluated at the UNROUNDED float64 value would sit 3e-4 rad away in the high-frequency channels)
    time = torch.linspace(0.2, 1.0, nT, dtype=torch.float64).float().double()
    return kw, cfg, P, keys, query, Ts, time


def oracle_run(kw, P, keys, query, Ts, time, dtype):
    ocfg = R.config_from_kwargs(kw)
    Pd = R.cast_params(P, dtype)
    okeys = [R.FeaturedPoints(k.x.to(dtype), k.f.to(dtype), k.b, None if k.w is None else k.w.to(dtype)) for k in keys]
    oq = R.FeaturedPoints(query.x.to(dtype), query.f.to(dtype), query.b, query.w.to(dtype))
    dbg = R.Debug()
    ang, lin = R.score_head_forward(ocfg, Pd, Ts.to(dtype), okeys, oq, time.float().to(dtype), dbg)      # (the time as the C ABI receives it: float32)
    return ang, lin, dbg, ocfg


def gpu_run(kw, P, keys, query, Ts, time, debug=True, half=False):
    dev = torch.device('cuda:0')
    head = ScoreModelHead(**kw)
    head.load_state_dict(P)
    head.to(dev)
    if half:
        head.half()          # the reference's half_precision switch (agent.py:50-51)
    gkeys = [FeaturedPoints(k.x.to(dev), k.f.to(dev), k.b.to(dev), None if k.w is None else k.w.to(dev)) for k in keys]
    gq = FeaturedPoints(query.x.to(dev), query.f.to(dev), query.b.to(dev), query.w.to(dev))
    head.set_key_clouds(gkeys)
    head.set_query(gq)
    if debug:
        head.debug_enable(True)
    ang, lin = head(Ts.to(dev, torch.float32), gkeys, gq, time.to(dev, torch.float32))
    torch.cuda.synchronize()
    return head, ang.cpu(), lin.cpu()


def stage_report(lmax=2, nT=6, n_scene=512, n_grasp=100, verbose=True, half=False, **kwargs):
    return stage_report_case(*build_case(lmax, nT, n_scene, n_grasp, **kwargs), verbose=verbose, half=half)


def stage_report_case(kw, cfg, P, keys, query, Ts, time, verbose=True, half=False):
    """every stage of the HIP path (debug buffers of the C ABI) against the fp64 oracle on one case"""
    nT = len(Ts)
    ang64, lin64, d64, ocfg = oracle_run(kw, P, keys, query, Ts, time, torch.float64)
    ang32, lin32, d32, _ = oracle_run(kw, P, keys, query, Ts, time, torch.float32)
    head, ang, lin = gpu_run(kw, P, keys, query, Ts, time, half=half)
    muls = cfg.muls
    D = cfg.dim
    DK = sum(KERNEL_MULS[l] * (2 * l + 1) for l in range(len(muls)))          # feature width of the kernels' buffers (= D up to lmax 2)
    nQ = len(query.x)
    Nd = nT * nQ
    perm = int_to_ref_perm(muls)
    rep = {}
    st = head.stats()
    rep['edges_gpu'] = st['n_edges']
    rep['edges_oracle'] = d64['n_edges_per_scale']
    rep['msg'] = rel(head.debug_buffer('msg').reshape(-1, DK)[:, kernel_ref_to_true_perm(muls)], d64['msg_src'])
    rep['qpos'] = rel(head.debug_buffer('qpos').reshape(nT, nQ, 3), d64['x_t'])
    pose = head.debug_buffer('pose').reshape(nT, -1)
    # Wigner D^1 against the oracle's rotated features: f_t(l=1 block) = D^1 f
    # edges: map (scale, dst, src) -> row
    es = head.debug_buffer('edge_src', torch.int32).long()
    ed = head.debug_buffer('edge_dst', torch.int32).long()
    ti = head.debug_buffer('tile_info', torch.int32)
    E = int(ti[16 + cfg.n_scales])
    es, ed = es[:E], ed[:E]
    nkeys = sum(len(k.x) for k in keys)
    key_g = ed * nkeys + es
    key_o = d64['edge_dst'] * nkeys + d64['edge_src']
    rep['edge_set_equal'] = bool(E == len(key_o) and torch.equal(torch.sort(key_g).values, torch.sort(key_o).values))
    if rep['edge_set_equal']:
        og = torch.argsort(key_g)
        oo = torch.argsort(key_o)
        WN = d64['dtp_weight'].shape[1]
        wperm = dtp_weight_perm(muls)
        assert len(wperm) == WN
        w = head.debug_buffer('dbg_w').reshape(E, -1)[:, wperm]
        rep['dtp_weight'] = rel(w[og], d64['dtp_weight'][oo])
        rep['dtp_weight_o32'] = rel(d32['dtp_weight'][torch.argsort(d32['edge_dst'] * nkeys + d32['edge_src'])], d64['dtp_weight'][oo])
        eo = head.debug_buffer('edge_out').reshape(-1, DK + 4)[:E]
        val_ref = eo[:, :DK][:, perm]
        H = cfg.num_heads
        irreps_head = [(m // H, l) for l, m in enumerate(muls)]
        oval = R.heads2vec(d64['value'], irreps_head)
        rep['value'] = rel(val_ref[og], oval[oo])
        off = 0
        for l, m in enumerate(muls):      # per irreps block, each relative to its own maximum
            n = m * (2 * l + 1)
            rep[f'value_l{l}'] = rel(val_ref[og][:, off:off + n], oval[oo][:, off:off + n])
            off += n
        rep['logits'] = rel(eo[:, DK:][og], d64['log_alpha'][oo])
    z = head.debug_buffer('z').reshape(Nd, DK)[:, perm]
    rep['attn'] = rel(z, d64['attn'])
    rep['attn_o32'] = rel(d32['attn'], d64['attn'])
    for nm in ('emb', 'field'):          # proj output and the field after post-norm + FFN, per irreps block
        g = head.debug_buffer(nm).reshape(Nd, DK)[:, perm]
        off = 0
        for l, m in enumerate(muls):
            nn = m * (2 * l + 1)
            rep[f'{nm}_l{l}'] = rel(g[:, off:off + nn], d64[nm][:, off:off + nn])
            rep[f'o32_{nm}_l{l}'] = rel(d32[nm][:, off:off + nn], d64[nm][:, off:off + nn])      # the fp32 restatement's own distance (diagnostic)
            off += nn
    no = head.debug_buffer('node_out').reshape(nT, nQ, 8)
    w = query.w.double()
    lin_q = d64['lin_vel_q'] * w[None, :, None]
    rep['node_lin'] = rel(no[:, :, :3], lin_q)
    scale = float(max(ang64.abs().max(), lin64.abs().max()))
    rep['final_ang'] = float((ang.double() - ang64).abs().max() / scale)
    rep['final_lin'] = float((lin.double() - lin64).abs().max() / scale)
    rep['final_o32_ang'] = float((ang32.double() - ang64).abs().max() / scale)
    rep['final_o32_lin'] = float((lin32.double() - lin64).abs().max() / scale)
    if verbose:
        for k, v in rep.items():
            print(f"{k:18s} {v}")
    return rep


if __name__ == '__main__':
    a = [int(x) for x in sys.argv[1:]]
    stage_report(*(a + [2, 6, 512, 100][len(a):]))
