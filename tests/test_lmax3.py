"""lmax 3 (irreps 64x0e+32x1e+16x2e+8x3e, SH up to 3e) through the whole path -- BASELINE config 5's degree, SURVEY 8(a) lmax-3 row
(D = 296, weight_numel 800, DTP output 3 488, M_edge 338 304).  No reference config uses it; the reference code is irreps-generic
(multiscale_tensor_field.py:22-190, equiformer/tensor_product_rescale.py:352-382 with the output filter at :368, graph_parser.py:135), and so is
the oracle.  The kernels run the 8x3e block as a zero-padded 16x3e chunk (csrc/dedf_net.h::mul_of / pad_pos, dedf_pack.h::pad_params): the C ABI
and these tests speak the reference's TRUE shapes.

CPU: schema, the l = 3 constants (SH equivariance under the reference's own Wigner-D recipe pins the +-m pair signs of SURVEY 8(c) item 2
together with the w3j(.,.,3) equivariance of tests/test_so3.py), the zero-padded embedding of the UNet layers proven on the oracle.
GPU: the HIP path against the fp64 oracle at the same bars as lmax 2."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

import stage_check as SC
from diffusion_edf_amd import _lib, params, so3, synthetic
from diffusion_edf_amd.gnn_data import FeaturedPoints
from oracle import graph_oracle as GO
from oracle import restatement as R
from oracle import so3_oracle as oso3
from oracle import unet_oracle as U

TOL = 1e-4
IRREPS3 = [(64, 0), (32, 1), (16, 2), (8, 3)]
NARROW3 = [(32, 0), (16, 1), (8, 2), (4, 3)]
SH3 = [(1, 0), (1, 1), (1, 2), (1, 3)]


# ---- CPU ----------------------------------------------------------------------------------------------------------------------------

def test_sizes_of_the_survey_lmax3_row():
    cfg = params.HeadConfig.from_kwargs(synthetic.score_head_kwargs(3))
    assert cfg.dim == 296 and cfg.lmax == 3 and cfg.lmax_sh == 3
    paths = params.dtp_paths(cfg.irreps, [0, 1, 2, 3], [1] * 4, [0, 1, 2, 3])
    assert len(paths) == 34 and sum(p[3] for p in paths) == 800
    by = params.dtp_sorted_out(paths)
    k = [sum(paths[p][3] for p in by[l]) for l in range(4)]
    assert k == [120, 224, 248, 208] and sum(kk * (2 * l + 1) for l, kk in enumerate(k)) == 3488
    # algorithmic MAC per edge (dense-CG convention): pre-linear + radial MLP + 2 depth-wise TPs + lin + sep_alpha + lin2
    dense_cg = sum(m1 * (2 * l1 + 1) * (2 * l2 + 1) * (2 * l3 + 1) for l1, l2, l3, m1, _ in paths)
    lin_out = [64 + 32 + 16 + 8, 32, 16, 8]
    m_edge = 128 * 128 + (128 * 128 + 128 * 64 + 64 * 800) + 2 * dense_cg + sum(kk * o * (2 * l + 1) for l, (kk, o) in enumerate(zip(k, lin_out))) \
        + 120 * 64 + sum(kk * o * (2 * l + 1) for l, (kk, o) in enumerate(zip(k, [64, 32, 16, 8])))
    assert m_edge == 338_304


def test_l3_harmonics_pair_signs_are_pinned_by_the_reference_wigner_recipe():
    """Y_3(R p) = D^3(R) Y_3(p) with D^3 = X(a) J_3 X(b) J_3 X(c) of reference wigner.py:44-81 and the YXY angles of transforms.py -- a sign
    flip of ONE member of a +-m pair breaks it; the two independent derivations (product generator, oracle) agree to 1e-12"""
    rng = np.random.default_rng(3)
    p = rng.normal(size=(50, 3))
    for a, b, c in rng.uniform(0.3, 2.8, size=(5, 3)):
        ry = lambda t: np.array([[np.cos(t), 0, np.sin(t)], [0, 1, 0], [-np.sin(t), 0, np.cos(t)]])
        rx = lambda t: np.array([[1, 0, 0], [0, np.cos(t), -np.sin(t)], [0, np.sin(t), np.cos(t)]])
        Rm = ry(a) @ rx(b) @ ry(c)
        D = so3.wigner_D(3, a, b, c)[0]
        Y, YR = so3.spherical_harmonics(3, p), so3.spherical_harmonics(3, p @ Rm.T)
        assert np.abs(YR - Y @ D.T).max() < 1e-12
        for m in range(7):                                     # flipping any single component's sign is detected
            if m == 3:
                continue
            S = np.eye(7); S[m, m] = -1
            assert np.abs(YR @ S - (Y @ S) @ D.T).max() > 1e-3
    assert np.abs(so3.spherical_harmonics(3, p) - oso3.sh(3, p)).max() < 1e-12
    for t in [(1, 2, 3), (2, 2, 3), (2, 1, 3), (3, 3, 3), (3, 1, 2), (3, 3, 0), (1, 3, 3), (0, 3, 3)]:
        Cw = so3.wigner_3j(*t)
        Ds = [so3.wigner_D(l, 0.7, 1.3, 2.1)[0] for l in t]
        assert np.abs(np.einsum('ijk,ai,bj,ck->abc', Cw, *Ds) - Cw).max() < 1e-12 and np.abs(Cw - oso3.w3j(*t)).max() < 1e-10


def test_library_schema_and_packing_at_lmax3(built_lib):
    for kwf in (synthetic.score_head_kwargs, synthetic.ebm_head_kwargs):
        cfg = params.HeadConfig.from_kwargs(kwf(3))
        cc = _lib.make_config(cfg, -1)
        assert _lib.param_names(cc) == [(n, int(np.prod(s))) for n, s, _, _ in params.param_spec(cfg)]       # the reference's TRUE shapes
        blob = _lib.pack_params(cc, params.init_params(cfg, 2, True))
        h = C.c_void_p()
        assert built_lib.dedf_create(C.byref(cc), blob.ctypes.data_as(C.POINTER(C.c_float)), blob.size, C.byref(h)) == _lib.OK
        built_lib.dedf_destroy(h)
    # UNet-layer handles take the KERNEL shapes (unet_pad builds them): 64x0e+32x1e+16x2e+16x3e, FFN hidden 192 / 96 / 48 / 32
    from diffusion_edf_amd import unet_pad as UP
    cc = _lib.make_unet_layer_config(15.0, -1, muls=tuple(UP.WIDE3), valid=(64, 32, 16, 8))
    spec_w = params.unet_layer_param_spec([(m, l) for l, m in enumerate(UP.WIDE3)], [64, 32, 32], mid_muls=UP.WIDE_HID)
    assert _lib.param_names(cc) == [(n, int(np.prod(s))) for n, s, _, _ in spec_w]
    bad = _lib.make_unet_layer_config(15.0, -1, muls=tuple(UP.WIDE3))                 # lmax 3 must name the true 3e multiplicity
    assert built_lib.dedf_param_count(C.byref(bad)) == -1


@pytest.mark.parametrize("muls,muls_src,fc", [([64, 32, 16, 8], [64, 32, 16, 8], [64, 32, 32]),      # the wide levels: 8x3e itself is padded
                                               ([32, 16, 8, 4], [32, 16, 8, 4], [32, 16, 16]),        # the fine levels
                                               ([64, 32, 16, 8], [32, 16, 8, 4], [64, 32, 32]),        # pool layer between widths
                                               ([32, 16, 8, 4], [64, 32, 16, 8], [64, 32, 32])])       # unpool layer
def test_lmax3_layer_equals_its_zero_padded_kernel_shape_embedding(muls, muls_src, fc):
    """the embedding the lmax-3 kernels rely on, proven on the oracle in fp64: a layer with 8x3e (or 4x3e) is EXACTLY the
    64x0e+32x1e+16x2e+16x3e layer with FFN hidden 192/96/48/32, zero-padded parameters, true channels placed per head, LayerNorm statistics
    over the true channels"""
    from diffusion_edf_amd import unet_pad as UP
    irr = [(m, l) for l, m in enumerate(muls)]
    irr_s = [(m, l) for l, m in enumerate(muls_src)]
    P = R.cast_params(params.init_from_spec(params.unet_layer_param_spec(irr, fc, irreps_src=irr_s), seed=9, randomize_all=True), torch.float64)
    g = torch.Generator().manual_seed(1)
    gg = torch.Generator().manual_seed(0)
    x_src = torch.rand(60, 3, generator=gg, dtype=torch.float64) * 12.0
    x_dst = x_src[torch.randperm(60, generator=gg)[:14]] + 0.3 * torch.randn(14, 3, generator=gg, dtype=torch.float64)
    ed, es = R.radius_bipartite(x_src, x_dst, 9.0)
    f_src = torch.randn(len(x_src), sum(m * (2 * l + 1) for l, m in enumerate(muls_src)), generator=g, dtype=torch.float64)
    f_dst = torch.randn(len(x_dst), sum(m * (2 * l + 1) for l, m in enumerate(muls)), generator=g, dtype=torch.float64)
    true = U.layer_forward(U.LayerConfig(irr, SH3, 4, fc, radius=9.0, irreps_src=irr_s), P, x_src, f_src, x_dst, f_dst, es, ed)
    Q = R.cast_params(UP.expand_layer_params(P, muls, fc, muls_src), torch.float64)
    wide_irr = [(m, l) for l, m in enumerate(UP.WIDE3)]
    spec_w = params.unet_layer_param_spec(wide_irr, [64, 32, 32], mid_muls=UP.WIDE_HID)
    assert {n for n, _, _, _ in spec_w} == set(Q) and all(Q[n].numel() == int(np.prod(sh)) for n, sh, _, _ in spec_w)
    Q = {n: Q[n].reshape(sh) for n, sh, _, _ in spec_w}
    wide_cfg = U.LayerConfig(wide_irr, SH3, 4, [64, 32, 32], radius=9.0, valid=muls, fc_valid=fc, mid_muls=UP.WIDE_HID)
    wide = U.layer_forward(wide_cfg, Q, x_src, UP.pad_features(f_src, muls_src), x_dst, UP.pad_features(f_dst, muls), es, ed)
    back = UP.unpad_features(wide, muls)
    assert back.shape == true.shape and float((back - true).abs().max()) < 1e-11 * float(true.abs().max())
    mask_true = UP.pad_features(torch.ones(1, true.shape[1], dtype=torch.float64), muls)[0] > 0
    assert wide.shape[1] == 352 and float(wide[:, ~mask_true].abs().max()) == 0.0


@pytest.mark.parametrize("muls,muls_src,fc", [([64, 32, 16, 8], [64, 32, 16, 8], [64, 32, 32]), ([32, 16, 8, 4], [32, 16, 8, 4], [32, 16, 16]),
                                               ([64, 32, 16, 8], [32, 16, 8, 4], [64, 32, 32]), ([32, 16, 8, 4], [64, 32, 16, 8], [64, 32, 32])])
def test_lmax3_layer_blob_with_misplaced_3e_channels_is_refused(built_lib, muls, muls_src, fc):
    """the lmax-3 layer kernels skip the channels p with p % 4 >= 2 (narrow instantiations: >= 1) of the 16x3e block, so a C-ABI
    caller that zero-pads 8x3e differently from unet_pad.place would get silently wrong features.  dedf_create checks the placement on the tensors
    whose l = 3 block is addressable by name: the blobs unet_pad builds are accepted (wide, narrow and both mixed layers), the same blob with a true
    3e channel moved to a skipped position is DEDF_ERR_INVALID."""
    from diffusion_edf_amd import unet_pad as UP
    irr, irr_s = [(m, l) for l, m in enumerate(muls)], [(m, l) for l, m in enumerate(muls_src)]
    P = params.init_from_spec(params.unet_layer_param_spec(irr, fc, irreps_src=irr_s), seed=9, randomize_all=True)
    Q = UP.expand_layer_params(P, muls, fc, muls_src)
    nw = list(muls) == list(muls_src) == list(UP.NARROW3)
    cc = _lib.make_unet_layer_config(9.0, -1, UP.WIDE_FC, tuple(UP.WIDE3), 4, valid=tuple(muls), fc_valid=tuple(fc) if list(fc) != list(UP.WIDE_FC) else None, narrow=nw)
    def create(state):
        blob = _lib.pack_params(cc, state)
        h = C.c_void_p()
        rc = built_lib.dedf_create(C.byref(cc), blob.ctypes.data_as(C.POINTER(C.c_float)), blob.size, C.byref(h))
        if rc == _lib.OK:
            built_lib.dedf_destroy(h)
        return rc
    assert create(Q) == _lib.OK
    for name in ("gnn.linear_src.tp.weight", "gnn.ga.proj.tp.weight", "gnn.norm_2.affine_weight"):
        bad = {k: v.clone() for k, v in Q.items()}
        w = bad[name].reshape(-1)
        if name.endswith("affine_weight"):
            blk = w[64 + 32 + 16:]                       # the 16 per-channel weights of the 3e block
            assert float(blk[0]) != 0.0 and float(blk[3]) == 0.0
            blk[3] = blk[0]; blk[0] = 0.0                # channel 0 -> position 3 (p % 4 = 3: skipped by every lmax-3 kernel)
        else:
            blk = w[64 * 64 + 32 * 32 + 16 * 16:].reshape(16, 16)       # [in channel][out channel] of the l = 3 block
            assert float(blk[0].abs().max()) > 0.0 and float(blk[:, 3].abs().max()) == 0.0
            blk[:, 3] = blk[:, 0]; blk[:, 0] = 0.0       # output channel 0 -> position 3
        assert create(bad) == _lib.ERR_INVALID, name


# ---- GPU: score head ------------------------------------------------------------------------------------------------------------------

def _gpu_head(kw, P, dev, cls=None):
    from diffusion_edf_amd.score_head import ScoreModelHead
    head = (cls or ScoreModelHead)(**kw)
    head.load_state_dict(P)
    return head.to(dev)


def _check(rep):
    assert rep['edges_gpu'] == rep['edges_oracle'] and rep['edge_set_equal']
    print("stage errors lmax3:", {k: f"{v:.1e}" for k, v in rep.items() if isinstance(v, float)})
    assert rep['final_ang'] < TOL and rep['final_lin'] < TOL, rep
    for k in ('msg', 'qpos', 'dtp_weight', 'value', 'attn', 'node_lin'):
        assert rep[k] < 1e-4, (k, rep[k])
    for k, v in rep.items():
        if k.startswith(('value_l', 'emb_l', 'field_l')):
            assert v < 1e-4, (k, v)
    assert 'value_l3' in rep and 'field_l3' in rep


@pytest.mark.gpu
def test_score_parity_fake_input_style_lmax3():
    """sizes of ScoreModelHead._get_fake_input (score_head.py:220-246), every stage of the HIP path per irreps block (incl. the 3e block)"""
    _check(SC.stage_report(lmax=3, nT=5, n_scene=512, n_grasp=100, verbose=False))
    # ragged / empty neighbourhoods and the notebook's identity-quaternion seed (YXY quirk: D^3(R_y(pi)) on the features, identity on the positions)
    _check(SC.stage_report(lmax=3, nT=7, n_scene=300, n_grasp=40, verbose=False, radii=(3.5, 5., 6.5, 8.), near=False))


@pytest.mark.gpu
def test_fake_input_runs_through_warmup_lmax3():
    from diffusion_edf_amd.score_head import ScoreModelHead
    head = ScoreModelHead(**synthetic.score_head_kwargs(3)).to('cuda:0')
    Ts, keys, query, time = head._get_fake_input()
    assert keys[0].f.shape == (100, 296) and query.f.shape == (10, 296)
    ang, lin = head.warmup(Ts, keys, query, time)
    assert ang.shape == (5, 3) and torch.isfinite(ang).all() and torch.isfinite(lin).all()


@pytest.mark.gpu
def test_full_size_c2_lmax3_anchored_on_the_oracle_through_pose_independence():
    """`bench.py --lmax 3` inputs (820/164/33/7 keys, 103 queries, 1000 poses): an 8-pose subset against the fp64 oracle at the full scene size,
    and the same poses inside the 1000-pose batch"""
    import bench
    dev = torch.device('cuda:0')
    kw, cfg, P, keys, query, Ts = bench.build_inputs(3, 4096, 1024, 1000, 0, dev)
    assert cfg.dim == 296 and [len(k.x) for k in keys] == [820, 164, 33, 7] and len(query.x) == 103
    head = _gpu_head(kw, P, dev)
    sel = torch.tensor([0, 1, 99, 333, 512, 777, 998, 999], device=dev)
    t_all = torch.full((1000,), 0.5, device=dev)
    ang_all, lin_all = head(Ts.float(), keys, query, t_all)
    assert head.stats()['n_edges_total'] > 1_000_000 and not head.stats()['overflow']
    assert max(head.stats()['rtab_err']) > 0.0          # one time for every pose: forward took the radial table behind its launch gate (and the guard ran)
    head.set_radial_table("always")                     # (the small subset would evaluate per edge on its own: the table differs from that by ~3e-6)
    ang_s, lin_s = head(Ts[sel].float(), keys, query, t_all[:len(sel)])
    scale = float(max(ang_all.abs().max(), lin_all.abs().max()))
    assert float((ang_all[sel] - ang_s).abs().max()) / scale < 2e-6 and float((lin_all[sel] - lin_s).abs().max()) / scale < 2e-6
    ocfg = R.config_from_kwargs(kw)
    ok = [R.FeaturedPoints(k.x.cpu().double(), k.f.cpu().double(), k.b.cpu()) for k in keys]
    oq = R.FeaturedPoints(query.x.cpu().double(), query.f.cpu().double(), query.b.cpu(), query.w.cpu().double())
    d = R.Debug()
    ang64, lin64 = R.score_head_forward(ocfg, R.cast_params(P, torch.float64), Ts[sel].cpu(), ok, oq, torch.full((len(sel),), 0.5, dtype=torch.float64), d)
    assert sum(d['n_edges_per_scale']) > 8_000
    s64 = float(max(ang64.abs().max(), lin64.abs().max()))
    assert float((ang_all[sel].cpu().double() - ang64).abs().max()) / s64 < TOL and float((lin_all[sel].cpu().double() - lin64).abs().max()) / s64 < TOL


@pytest.mark.gpu
def test_sampler_lmax3_against_the_oracle_loop_with_and_without_the_radial_table():
    from diffusion_edf_amd.score_model_base import ScoreModelBase
    dev = torch.device('cuda:0')
    kw, cfg, P, keys, query, Ts, _ = SC.build_case(3, 6, 512, 60)
    head = _gpu_head(kw, P, dev)
    gk = [FeaturedPoints(k.x.to(dev), k.f.to(dev), k.b.to(dev)) for k in keys]
    gq = FeaturedPoints(query.x.to(dev), query.f.to(dev), query.b.to(dev), query.w.to(dev))
    noise = torch.randn(3, 2, len(Ts), 3, generator=torch.Generator().manual_seed(0), dtype=torch.float64)
    ok = [R.FeaturedPoints(k.x, k.f, k.b, None) for k in keys]
    ref = R.sample(R.config_from_kwargs(kw), P, Ts, ok, R.FeaturedPoints(query.x, query.f, query.b, query.w), [[0.8, 0.3]], [3], [0.04],
                   temperatures=1.0, noise=noise)
    outs = {}
    for on in ("always", False):
        head.set_radial_table(on)
        outs[on] = ScoreModelBase(head).sample(Ts.to(dev), gk, gq, [[0.8, 0.3]], [3], [0.04], temperatures=1.0, noise=noise).cpu()
        print(f"lmax3 sample, table={on}: {float((outs[on] - ref).abs().max()):.2e}")
        assert outs[on].shape == ref.shape and float((outs[on] - ref).abs().max()) < 2e-5, (on, float((outs[on] - ref).abs().max()))
    # one noise-free step: displacement difference = score difference; the table path stays within 1e-5 of the per-edge path
    d = {}
    for on in ("always", False):
        head.set_radial_table(on)
        o = ScoreModelBase(head).sample(Ts.to(dev), gk, gq, [[0.5, 0.5]], [1], [0.04], temperatures=0.0).cpu()
        d[on] = (o[1] - o[0])[:, 4:]
    scale = float(d[False].abs().max())
    dev_ = float((d["always"] - d[False]).abs().max()) / scale
    assert 0.0 < dev_ < 1e-5, dev_


@pytest.mark.gpu
def test_ebm_energy_parity_lmax3():
    from diffusion_edf_amd.score_head import EbmScoreModelHead
    dev = torch.device('cuda:0')
    kw = synthetic.ebm_head_kwargs(3)
    cfg = params.HeadConfig.from_kwargs(kw)
    P = params.init_params(cfg, seed=4, randomize_all=True)
    keys = synthetic.make_key_clouds(cfg, 600, seed=2)
    query = synthetic.make_query(cfg, 80, seed=2)
    Ts = synthetic.make_poses(9, seed=5, near_object=True)
    hk = {k: v for k, v in kw.items() if k != 'ebm'}
    head = _gpu_head(hk, P, dev, cls=EbmScoreModelHead)
    gk = [FeaturedPoints(k.x.to(dev), k.f.to(dev), k.b.to(dev)) for k in keys]
    gq = FeaturedPoints(query.x.to(dev), query.f.to(dev), query.b.to(dev), query.w.to(dev))
    e = head.compute_energy(Ts.float().to(dev), gk, gq, torch.zeros(len(Ts), device=dev)).cpu().double()
    ok = [R.FeaturedPoints(k.x.double(), k.f.double(), k.b) for k in keys]
    oq = R.FeaturedPoints(query.x.double(), query.f.double(), query.b, query.w.double())
    e64 = R.compute_energy(R.config_from_kwargs(kw), R.cast_params(P, torch.float64), Ts, ok, oq, torch.zeros(len(Ts), dtype=torch.float64))
    assert float((e - e64).abs().max()) / float(e64.abs().max()) < TOL
    order = torch.argsort(e64)                    # the ranking agent.py:172-173 sorts by: the oracle's order is non-decreasing on the GPU values too (ties: isolated poses)
    assert bool((e[order][1:] - e[order][:-1] > -1e-4 * float(e64.abs().max())).all())


@pytest.mark.gpu
def test_half_precision_mode_lmax3_score_head_critic_and_field():
    """`model.half()` (reference agent.py:50-51) at lmax 3: the score head, the EBM critic and the context-free [64, 32, 32] field with every GEMM
    as ONE fp16 MFMA product (k_edge<3, ., true> / k_node<3, ., true>), against the fp64 oracle at the half mode's stated 5e-3 -- and really
    different from the full-precision mode"""
    from diffusion_edf_amd.score_head import EbmScoreModelHead
    rep = SC.stage_report(lmax=3, nT=5, n_scene=512, n_grasp=100, verbose=False, half=True)
    assert rep['edges_gpu'] == rep['edges_oracle'] and rep['edge_set_equal']
    assert rep['final_ang'] < 5e-3 and rep['final_lin'] < 5e-3, rep
    assert max(rep['final_ang'], rep['final_lin']) > 3e-5, rep
    # the critic
    dev = torch.device('cuda:0')
    kw = synthetic.ebm_head_kwargs(3)
    cfg = params.HeadConfig.from_kwargs(kw)
    P = params.init_params(cfg, seed=4, randomize_all=True)
    keys = synthetic.make_key_clouds(cfg, 600, seed=2)
    query = synthetic.make_query(cfg, 80, seed=2)
    Ts = synthetic.make_poses(9, seed=5, near_object=True)
    head = _gpu_head({k: v for k, v in kw.items() if k != 'ebm'}, P, dev, cls=EbmScoreModelHead)
    head.half()
    gk = [FeaturedPoints(k.x.to(dev), k.f.to(dev), k.b.to(dev)) for k in keys]
    gq = FeaturedPoints(query.x.to(dev), query.f.to(dev), query.b.to(dev), query.w.to(dev))
    e = head.compute_energy(Ts.float().to(dev), gk, gq, torch.zeros(len(Ts), device=dev)).cpu().double()
    ok = [R.FeaturedPoints(k.x.double(), k.f.double(), k.b) for k in keys]
    oq = R.FeaturedPoints(query.x.double(), query.f.double(), query.b, query.w.double())
    e64 = R.compute_energy(R.config_from_kwargs(kw), R.cast_params(P, torch.float64), Ts, ok, oq, torch.zeros(len(Ts), dtype=torch.float64))
    err = float((e - e64).abs().max()) / float(e64.abs().max())
    assert 3e-6 < err < 5e-3, err
    # the field (KeypointExtractor.tensor_field's shape) at both degrees: half against full precision of the same module
    from diffusion_edf_amd.keypoint_extractor import MultiscaleTensorField
    for lmax in (2, 3):
        irr = '+'.join(['64x0e', '32x1e', '16x2e', '8x3e'][:lmax + 1])
        sh = '+'.join(['1x0e', '1x1e', '1x2e', '1x3e'][:lmax + 1])
        tf = MultiscaleTensorField(irreps_input=irr, irreps_output=irr, irreps_sh=sh, num_heads=4, fc_neurons=[-1, 32, 32], length_emb_dim=64,
                                   irreps_query=None, r_cluster_multiscale=[5.0, 10.0], edge_context_emb_dim=None, n_scales=2, init_seed=5).to(dev)
        kc = params.HeadConfig.from_kwargs(synthetic.score_head_kwargs(lmax, radii=(5., 10.)))
        kk = synthetic.make_key_clouds(kc, 1500, seed=3)
        gkk = [FeaturedPoints(k.x.to(dev), k.f.to(dev), k.b.to(dev)) for k in kk]
        qp = FeaturedPoints(x=gkk[0].x[:200] + 0.3, f=torch.empty(200, 0, device=dev), b=torch.zeros(200, dtype=torch.long, device=dev), w=None)
        full = tf(qp, gkk).f
        tf.half()
        hf = tf(qp, gkk).f
        errf = float((hf - full).abs().max()) / float(full.abs().max())
        assert 1e-6 < errf < 5e-3, (lmax, errf)


# ---- GPU: the UNet feature extractor at lmax 3 (BASELINE config 5) ------------------------------------------------------------------

def _check_layer3(x_src, x_dst, es, ed, radius, seed, muls, muls_src=None, fc=(64, 32, 32), half=False, tol=1e-4):
    from diffusion_edf_amd.unet import UnetLayer
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(seed)
    muls, muls_src, fc = list(muls), list(muls if muls_src is None else muls_src), list(fc)
    irr, irr_s = [(m, l) for l, m in enumerate(muls)], [(m, l) for l, m in enumerate(muls_src)]
    ds, dd = sum(m * (2 * l + 1) for l, m in enumerate(muls_src)), sum(m * (2 * l + 1) for l, m in enumerate(muls))
    f_src = torch.randn(len(x_src), ds, generator=g, dtype=torch.float64)
    f_dst = f_src[:len(x_dst)].clone() if (len(x_dst) == len(x_src) and ds == dd) else torch.randn(len(x_dst), dd, generator=g, dtype=torch.float64)
    P = params.init_from_spec(params.unet_layer_param_spec(irr, fc, irreps_src=irr_s), seed=seed, randomize_all=True)
    ref = U.layer_forward(U.LayerConfig(irr, SH3, 4, fc, radius=radius, irreps_src=irr_s), R.cast_params(P, torch.float64), x_src, f_src, x_dst, f_dst, es, ed)
    s_ = lambda mm: '+'.join(f"{m}x{l}e" for l, m in enumerate(mm))
    layer = UnetLayer(irreps=s_(muls), irreps_src=s_(muls_src), irreps_edge_attr='1x0e+1x1e+1x2e+1x3e', fc_neurons=fc, radius=radius)
    layer.load_state_dict(P)
    layer.to(dev)
    if half:
        layer.half()
    out = layer(x_src.float().to(dev), f_src.float().to(dev), x_dst.float().to(dev), f_dst.float().to(dev), es.to(dev), ed.to(dev)).cpu().double()
    assert out.shape == ref.shape
    o = 0
    for m, l in irr:
        d = m * (2 * l + 1)
        err = float((out[:, o:o + d] - ref[:, o:o + d]).abs().max()) / float(ref[:, o:o + d].abs().max())
        assert err < tol, (l, err)
        o += d


@pytest.mark.gpu
def test_unet_layers_lmax3_on_levels_of_a_16k_scene():
    """wide (64x0e+32x1e+16x2e+8x3e), fine (32x0e+16x1e+8x2e+4x3e), pool (fine source, wide block) and unpool (reversed graph) layers on graphs of
    the 16 384-point scene, fp32 mode at 1e-4 per irreps block and the half-precision GEMM mode (config 5's "fp16 MFMA path-weight GEMMs") at 5e-3"""
    from diffusion_edf_amd import connectivity as CN
    dev = torch.device('cuda:0')
    x = synthetic.make_scene(16384, seed=0).astype(np.float32)
    levels, radii = [], [3.0]
    for n in range(4):
        x = x[GO.fps(x, ratio=0.2, start=0)]
        levels.append(torch.from_numpy(x.copy()))
        if n:
            radii.append(radii[-1] / math.sqrt(0.2))
    z = lambda n: torch.zeros(n, dtype=torch.long, device=dev)
    xs = levels[2]
    rg = CN.RadiusGraph(r=radii[2], max_num_neighbors=1000)
    _, _, es, ed, _, _ = rg(xs.float().to(dev), torch.zeros(len(xs), 1, device=dev), z(len(xs)))
    _check_layer3(xs.float().double(), xs.float().double(), es.cpu(), ed.cpu(), radii[2], seed=3, muls=(64, 32, 16, 8))
    _check_layer3(xs.float().double(), xs.float().double(), es.cpu(), ed.cpu(), radii[2], seed=3, muls=(64, 32, 16, 8), half=True, tol=5e-3)
    x1 = levels[1].float()
    rg1 = CN.RadiusGraph(r=radii[1], max_num_neighbors=1000)
    _, _, es, ed, _, _ = rg1(x1.to(dev), torch.zeros(len(x1), 1, device=dev), z(len(x1)))
    _check_layer3(x1.double(), x1.double(), es.cpu(), ed.cpu(), radii[1], seed=4, muls=(32, 16, 8, 4), fc=(32, 16, 16))
    pool2 = CN.FpsPool(ratio=0.2, random_start=False, r=radii[2], max_num_neighbors=1000)
    _, x3, es, ed, _, _ = pool2(x1.to(dev), torch.zeros(len(x1), 1, device=dev), z(len(x1)))
    _check_layer3(x1.double(), x3.cpu().double(), es.cpu(), ed.cpu(), radii[2], seed=22, muls=(64, 32, 16, 8), muls_src=(32, 16, 8, 4))
    _check_layer3(x3.cpu().double(), x1.double(), ed.cpu(), es.cpu(), radii[2], seed=23, muls=(32, 16, 8, 4), muls_src=(64, 32, 16, 8))


def _randomized(module, seed):
    g = torch.Generator().manual_seed(seed)
    sd = module.state_dict()
    for k, v in sd.items():
        if k.endswith("parity_inversion.sign"):
            continue
        if "radial." in k or v.ndim == 0:
            sd[k] = v + 0.1 * torch.randn(v.shape, generator=g)
        elif "affine_weight" in k or (".net." in k and k.endswith(("1.weight", "4.weight"))):
            sd[k] = 1.0 + 0.2 * torch.randn(v.shape, generator=g)
        elif "bias" in k or k.endswith("offset"):
            sd[k] = v + 0.2 * torch.randn(v.shape, generator=g)
        else:
            sd[k] = v * (1.0 + 0.1 * torch.randn(v.shape, generator=g))
    module.load_state_dict(sd)
    return sd


_UNET_REF = {}


def unet_lmax3_reference(n_points):
    """The lmax-3 panda_lowres UNet with the weights of BASELINE config 5's key model (tests/test_config5.py::build_config5: seeded init, `_randomized`
    with seed 1) on the synthetic scene with config 5's features, and its fp64 restatement -- ONE ~140 s pass per scene size, shared by the
    UNet tests here and the config-5 chain test (which checks that its model's weights are these)."""
    from diffusion_edf_amd.so3 import parse_irreps
    from diffusion_edf_amd.unet import UnetFeatureExtractor
    if n_points not in _UNET_REF:
        m = UnetFeatureExtractor(**synthetic.unet_kwargs("panda_lowres_lmax3"), deterministic=True)
        sd = _randomized(m, seed=1)
        kw = m._ctor
        ocfg = U.UnetConfig(irreps_input=parse_irreps(kw["irreps_input"]), irreps_output=parse_irreps(kw["irreps_output"]),
                            irreps_emb=[parse_irreps(i) for i in kw["irreps_emb"]], fc_neurons=[list(f) for f in kw["fc_neurons"]],
                            n_layers=list(kw["n_layers"]), pool_ratio=list(kw["pool_ratio"]), radius=list(m.radius),
                            n_layers_midstream=kw["n_layers_midstream"], irreps_sh=SH3)
        x = torch.from_numpy(synthetic.make_scene(n_points, seed=0).astype(np.float32))
        f = torch.rand(n_points, 3, generator=torch.Generator().manual_seed(0))
        _UNET_REF[n_points] = dict(sd=sd, x=x, f=f, ref=U.unet_forward(ocfg, R.cast_params(sd, torch.float64), x, f.double()))
    return _UNET_REF[n_points]


@pytest.mark.gpu
@pytest.mark.parametrize("n_points,half", [(16384, False), (16384, True), (3000, True)])
def test_unet_feature_extractor_lmax3_matches_the_oracle(n_points, half):
    """BASELINE config 5 as written: the full UNet feature extractor on the 16 384-point scene at lmax 3 (levels 3277 / 656 / 132 / 27, 17 layers,
    SH up to 3e, parity-inverted up path) against the fp64 restatement -- coordinates bit-exact, features per output scale and irreps block within
    2e-5 of the block's magnitude (measured 5.5e-7 .. 2.3e-6, which is what the restatement itself shows when it runs in fp32:
    profiles/r04v_unet_lmax3_err.log); and its fp16-GEMM mode at the same size (measured 0.8 .. 4.1e-3, held to 8e-3) and on a smaller scene (5e-3)"""
    from diffusion_edf_amd.unet import UnetFeatureExtractor
    dev = torch.device("cuda:0")
    rf = unet_lmax3_reference(n_points)
    m = UnetFeatureExtractor(**synthetic.unet_kwargs("panda_lowres_lmax3"), deterministic=True)
    m.load_state_dict(rf["sd"])
    x, f, ref = rf["x"], rf["f"], rf["ref"]
    m.to(dev)
    if half:
        m.half()
    out = m(FeaturedPoints(x=x.to(dev), f=f.to(dev), b=torch.zeros(n_points, dtype=torch.long, device=dev), w=None))
    assert len(out) == len(ref) == 4
    tol = ((8e-3 if n_points > 3000 else 5e-3) if half else 2e-5)
    if n_points == 16384:
        assert [len(o.x) for o in out] == [3277, 656, 132, 27]
    for o, (xr, fr) in zip(out, ref):
        assert torch.equal(o.x.cpu(), xr) and o.f.shape == fr.shape == (len(xr), 296)
        got = o.f.cpu().double()
        off = 0
        for mul, l in IRREPS3:
            d = mul * (2 * l + 1)
            err = float((got[:, off:off + d] - fr[:, off:off + d]).abs().max()) / max(float(fr[:, off:off + d].abs().max()), 1e-3 * float(fr.abs().max()))
            print(f"unet lmax3 n={n_points} half={half} scale {len(xr)} l={l}: {err:.2e}")
            assert err < tol, (len(xr), l, err)
            off += d


@pytest.mark.gpu
def test_keypoint_extractor_lmax3_matches_the_oracle():
    """the place tasks' query model at lmax 3: UNet + FPS key points + tensor_field / weight_field (context-free MultiscaleTensorFields with the
    [64, 32, 32] radial MLP) + the weight head, against the fp64 restatement: coordinates bit-exact, features 2e-5, weights 2e-5 absolute (measured 9.5e-7 / 1.2e-7)"""
    from diffusion_edf_amd.keypoint_extractor import KeypointExtractor
    from diffusion_edf_amd.so3 import parse_irreps
    dev = torch.device("cuda:0")
    radii = (5.0, 10.0, 20.0, 40.0)
    qk = synthetic.keypoint_extractor_kwargs(radii, unet="panda_highres_lmax3")
    m = KeypointExtractor(**qk, deterministic=True)
    _randomized(m, seed=2)
    x = synthetic.make_scene(2500, seed=6).astype(np.float32)
    x = (x - x.mean(0)) * 0.5
    x[:, 2] += 14.0 - x[:, 2].min()
    x = torch.from_numpy(x.astype(np.float32))
    f = torch.rand(len(x), 3, generator=torch.Generator().manual_seed(0))
    kw = m.feature_extractor._ctor
    ucfg = U.UnetConfig(irreps_input=parse_irreps(kw["irreps_input"]), irreps_output=parse_irreps(kw["irreps_output"]),
                        irreps_emb=[parse_irreps(i) for i in kw["irreps_emb"]], fc_neurons=[list(v) for v in kw["fc_neurons"]],
                        n_layers=list(kw["n_layers"]), pool_ratio=list(kw["pool_ratio"]), radius=list(m.feature_extractor.radius),
                        n_layers_midstream=kw["n_layers_midstream"], irreps_sh=SH3)
    fcfg = R.Config(irreps=IRREPS3, irreps_sh=SH3, num_heads=4, fc_neurons=[64, 32, 32], length_emb_dim=64, r_cluster_multiscale=list(radii),
                    r_mincut_nonscalar_sh=0.01 * radii[0], length_enc_max_r=None, time_emb_mlp=[256, 128, 64], max_time=1.0, time_enc_n=10000.0,
                    lin_mult=1.0, ang_mult=1.0, edge_time_encoding=False)
    P = R.cast_params({k: v.cpu() for k, v in m.state_dict().items()}, torch.float64)
    xq, fq, wq = U.keypoint_extractor_forward(ucfg, fcfg, P, x, f.double(), 0.1, bbox=qk["keypoint_kwargs"]["bbox"])
    m.to(dev)
    out = m(FeaturedPoints(x=x.to(dev), f=f.to(dev), b=torch.zeros(len(x), dtype=torch.long, device=dev), w=None))
    assert torch.equal(out.x.cpu(), xq) and out.f.shape == fq.shape == (len(xq), 296) and len(xq) > 20
    print(f"keypoint extractor lmax3: f {float((out.f.cpu().double() - fq).abs().max()) / float(fq.abs().max()):.2e}  w {float((out.w.cpu().double() - wq).abs().max()):.2e}")
    assert float((out.f.cpu().double() - fq).abs().max()) < 2e-5 * float(fq.abs().max())
    assert float((out.w.cpu().double() - wq).abs().max()) < 2e-5 and float(wq.max() - wq.min()) > 1e-3
