"""Pins the CPU oracle: (i) golden vectors produced by the two importable reference modules (tests/golden/make_golden.py),
(ii) the invariants the reference is designed around: SE(3) bi-equivariance of the score, softmax
normalisation, continuity at the cut-offs, and the YXY signed-zero quirk."""
import os

import numpy as np
import pytest
import torch

from diffusion_edf_amd import params, synthetic
from oracle import restatement as R

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def gt():
    return np.load(os.path.join(G, "transforms.npz"))


@pytest.fixture(scope="module")
def gr():
    return np.load(os.path.join(G, "radial_func.npz"))


def T(a, dt=torch.float64):
    return torch.tensor(a, dtype=dt)


def test_transforms_golden(gt):
    q, p = T(gt["q"]), T(gt["p"])
    assert torch.equal(R.standardize_quaternion(q), T(gt["standardize"]))
    assert torch.allclose(R.quaternion_to_matrix(q), T(gt["to_matrix"]), atol=0, rtol=0)
    assert torch.equal(R.quaternion_apply(q, p), T(gt["apply"]))
    assert torch.equal(R.quaternion_invert(q), T(gt["invert"]))
    assert torch.equal(R.quaternion_raw_multiply(q, q.flip(0)), T(gt["raw_multiply"]))
    e64 = R.matrix_to_euler_yxy(R.quaternion_to_matrix(R.standardize_quaternion(q)))
    assert torch.equal(e64, T(gt["euler_yxy_f64"]))
    q32 = q.float()
    e32 = R.matrix_to_euler_yxy(R.quaternion_to_matrix(R.standardize_quaternion(q32)))
    assert torch.equal(e32, T(gt["euler_yxy_f32"], torch.float32))
    # the quirk of SURVEY §0: identity -> (0, 0, pi)
    assert gt["euler_yxy_f64"][0].tolist() == [0.0, 0.0, np.pi]


def test_radial_func_golden(gr):
    x = T(gr["ssc2_x"], torch.float32)
    assert torch.equal(R.soft_square_cutoff_2(x, (None, None, 4., 5.)), T(gr["ssc2_right"], torch.float32))
    assert gr["ssc2_right"].tolist() == [1.0] * 9 + [0.6875, 0.0, 0.0, 0.0]
    x2 = T(gr["ssc2_x2"], torch.float32)
    assert torch.equal(R.soft_square_cutoff_2(x2, (0.2 * 0.3, 0.3, None, None)), T(gr["ssc2_left"], torch.float32))
    x3 = T(gr["ssc2_x3"], torch.float32)
    assert torch.equal(R.soft_square_cutoff_2(x3, (None, None, 0.8 * 20., 20.)), T(gr["ssc2_r20"], torch.float32))
    assert torch.equal(R.soft_step(torch.linspace(-0.5, 1.5, 41)), T(gr["soft_step"], torch.float32))
    d = T(gr["dist"], torch.float32)
    cfg = params.HeadConfig.from_kwargs(synthetic.score_head_kwargs(2))
    P = params.init_params(cfg, seed=2)
    for n, r in enumerate((5., 10., 20.)):
        out = R.gaussian_radial_basis(d, P, f"key_tensor_field.graph_parsers.{n}.length_enc", 64, r)
        assert torch.allclose(out, T(gr[f"grb_{int(r)}"], torch.float32), rtol=2e-6, atol=1e-7)
    assert torch.equal(R.sinusoidal_embedding(d * 3.0, 64, 100., 1000.), T(gr["sinus_len"], torch.float32))
    tt = T(gr["time"], torch.float32)
    assert torch.equal(R.sinusoidal_embedding(tt, 256, 1., 10000.), T(gr["sinus_time"], torch.float32))
    assert torch.equal(R.sinusoidal_embedding(tt.double(), 256, 1., 10000.), T(gr["sinus_time_f64"]))


# ---------------------------------------------------------------------------------------------------------------------

def _case(lmax, nT=5, n_scene=400, n_grasp=80, radii=(5., 10., 20., None), query_time_encoding=False):
    kw = synthetic.score_head_kwargs(lmax, radii=radii, query_time_encoding=query_time_encoding)
    cfgp = params.HeadConfig.from_kwargs(kw)
    cfg = R.config_from_kwargs(kw)
    P = params.init_params(cfgp, seed=2, randomize_all=True, dtype=torch.float64)
    keys = [R.FeaturedPoints(*k) for k in synthetic.make_key_clouds(cfgp, n_scene, dtype=torch.float64)]
    q = R.FeaturedPoints(*synthetic.make_query(cfgp, n_grasp, dtype=torch.float64))
    Ts = synthetic.make_poses(nT, near_object=True)
    time = torch.linspace(0.2, 0.9, nT, dtype=torch.float64)
    return cfg, P, keys, q, Ts, time


def _rand_q(seed):
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(4, generator=g, dtype=torch.float64)
    return q / q.norm()


@pytest.mark.parametrize("lmax", [1, 2])
def test_left_equivariance(lmax):
    """rotating/translating scene and poses together leaves the body-frame scores unchanged"""
    cfg, P, keys, q, Ts, time = _case(lmax)
    ang, lin = R.score_head_forward(cfg, P, Ts, keys, q, time)
    g, gt = _rand_q(3), torch.tensor([1., -2., 0.5], dtype=torch.float64)
    keys2 = [R.FeaturedPoints(x=R.quaternion_apply(g, k.x) + gt, f=R.transform_feature_quaternion(cfg.irreps, k.f, g[None])[0], b=k.b) for k in keys]
    Ts2 = torch.cat([R.quaternion_raw_multiply(g.expand(len(Ts), 4), Ts[:, :4]), R.quaternion_apply(g, Ts[:, 4:]) + gt], -1)
    ang2, lin2 = R.score_head_forward(cfg, P, Ts2, keys2, q, time)
    assert (ang2 - ang).abs().max() < 1e-11 and (lin2 - lin).abs().max() < 1e-11
    assert ang.abs().max() > 1e-3 and lin.abs().max() > 1e-3


@pytest.mark.parametrize("lmax", [1, 2])
def test_right_equivariance(lmax):
    """re-expressing the grasp cloud in a rotated gripper frame (x -> h^-1 x, f -> D(h^-1) f, T -> T h) rotates the
    body-frame scores by h^-1 (pure rotation h, so the orbital term transforms consistently)"""
    cfg, P, keys, q, Ts, time = _case(lmax)
    ang, lin = R.score_head_forward(cfg, P, Ts, keys, q, time)
    h = _rand_q(11)
    hinv = R.quaternion_invert(h)
    q2 = R.FeaturedPoints(x=R.quaternion_apply(hinv, q.x), f=R.transform_feature_quaternion(cfg.irreps, q.f, hinv[None])[0], b=q.b, w=q.w)
    Ts2 = torch.cat([R.quaternion_raw_multiply(Ts[:, :4], h.expand(len(Ts), 4)), Ts[:, 4:]], -1)
    ang2, lin2 = R.score_head_forward(cfg, P, Ts2, keys, q2, time)
    assert (ang2 - R.quaternion_apply(hinv, ang)).abs().max() < 1e-11
    assert (lin2 - R.quaternion_apply(hinv, lin)).abs().max() < 1e-11


def test_query_time_encoding_restatement():
    """query_time_encoding=True (score_head.py:168-173; gnn_block.py:109-130, 170-180, 205-206).  (a) The destination feature is a set of time
    scalars: both equivariances hold as before.  (b) It is read: another time changes the score through the destination side alone (edge time
    embedding held fixed by zeroing the time MLPs' last layers).  (c) With linear_dst and skip_1 zeroed the block is the plain one with a zero
    linear_src bias."""
    cfg, P, keys, q, Ts, time = _case(2, query_time_encoding=True)
    assert cfg.query_time_encoding and cfg.edge_time_encoding
    dbg = R.Debug()
    ang, lin = R.score_head_forward(cfg, P, Ts, keys, q, time, dbg)
    nQ = len(q.x)
    md = dbg['msg_dst'].reshape(len(Ts), nQ, -1)
    assert (md[:, 0:1] - md).abs().max() == 0 and (md[0, 0] - md[1, 0]).abs().max() > 1e-3       # one row per pose, different times differ
    assert md[..., 64:].abs().max() == 0                                                         # 0e channels only
    g, gt = _rand_q(3), torch.tensor([1., -2., 0.5], dtype=torch.float64)
    keys2 = [R.FeaturedPoints(x=R.quaternion_apply(g, k.x) + gt, f=R.transform_feature_quaternion(cfg.irreps, k.f, g[None])[0], b=k.b) for k in keys]
    Ts2 = torch.cat([R.quaternion_raw_multiply(g.expand(len(Ts), 4), Ts[:, :4]), R.quaternion_apply(g, Ts[:, 4:]) + gt], -1)
    ang2, lin2 = R.score_head_forward(cfg, P, Ts2, keys2, q, time)
    assert (ang2 - ang).abs().max() < 1e-11 and (lin2 - lin).abs().max() < 1e-11
    h = _rand_q(11)
    hinv = R.quaternion_invert(h)
    q2 = R.FeaturedPoints(x=R.quaternion_apply(hinv, q.x), f=R.transform_feature_quaternion(cfg.irreps, q.f, hinv[None])[0], b=q.b, w=q.w)
    Ts3 = torch.cat([R.quaternion_raw_multiply(Ts[:, :4], h.expand(len(Ts), 4)), Ts[:, 4:]], -1)
    ang3, lin3 = R.score_head_forward(cfg, P, Ts3, keys, q2, time)
    assert (ang3 - R.quaternion_apply(hinv, ang)).abs().max() < 1e-11 and (lin3 - R.quaternion_apply(hinv, lin)).abs().max() < 1e-11
    # (b)
    Pb = dict(P)
    for n in range(4):
        Pb[f"time_mlps_multiscale.{n}.2.weight"] = torch.zeros_like(P[f"time_mlps_multiscale.{n}.2.weight"])
    a1, _ = R.score_head_forward(cfg, Pb, Ts, keys, q, torch.full_like(time, 0.3))
    a2, _ = R.score_head_forward(cfg, Pb, Ts, keys, q, torch.full_like(time, 0.8))
    assert (a1 - a2).abs().max() > 1e-3 * a1.abs().max()
    # (c)
    blk = "key_tensor_field.gnn_block_init."
    Pc = dict(P)
    for k in ("linear_dst.tp.weight", "linear_dst.bias.0", "skip_1.skip.tp.weight", "skip_1.skip.bias.0"):
        Pc[blk + k] = torch.zeros_like(P[blk + k])
    ac, lc = R.score_head_forward(cfg, Pc, Ts, keys, q, time)
    cfg0 = R.config_from_kwargs(synthetic.score_head_kwargs(2))
    P0 = dict(Pc)
    P0[blk + "linear_src.bias.0"] = torch.zeros(64, dtype=torch.float64)
    a0, l0 = R.score_head_forward(cfg0, P0, Ts, keys, q, time)
    assert (ac - a0).abs().max() < 1e-13 and (lc - l0).abs().max() < 1e-13 and (ac - ang).abs().max() > 1e-3


def test_softmax_normalisation_and_empty_segments():
    cfg, P, keys, q, Ts, time = _case(2, radii=(3.5, 5., 6.5, 8.))      # high-res radii: far poses have no edges at all
    Ts = torch.cat([Ts, torch.tensor([[1., 0, 0, 0, 100., 100., 100.]], dtype=torch.float64)])
    time = torch.cat([time, time[-1:]])
    dbg = R.Debug()
    ang, lin = R.score_head_forward(cfg, P, Ts, keys, q, time, dbg)
    assert torch.isfinite(ang).all() and torch.isfinite(lin).all()
    Nd = len(Ts) * len(q.x)
    deg = torch.bincount(dbg['edge_dst'], minlength=Nd)
    assert (deg[-len(q.x):] == 0).all()                                   # the far pose has zero edges
    # attention output of edge-less nodes is exactly zero (scatter-sum of nothing)
    assert (dbg['attn'][-len(q.x):] == 0).all()


def test_continuity_at_cutoffs():
    """outputs are continuous when a key point crosses d = 0.8 r and d = r (soft cut-off acting on the logits)"""
    kw = synthetic.score_head_kwargs(2, radii=(5.,))
    cfgp = params.HeadConfig.from_kwargs(kw)
    cfg = R.config_from_kwargs(kw)
    P = params.init_params(cfgp, seed=2, randomize_all=True, dtype=torch.float64)
    g = torch.Generator().manual_seed(0)
    f = torch.randn(3, cfgp.dim, generator=g, dtype=torch.float64)
    qq = R.FeaturedPoints(x=torch.zeros(1, 3, dtype=torch.float64), f=torch.randn(1, cfgp.dim, generator=g, dtype=torch.float64),
                          b=torch.zeros(1, dtype=torch.long), w=torch.ones(1, dtype=torch.float64))
    Ts = torch.tensor([[1., 0, 0, 0, 0, 0, 0]], dtype=torch.float64)
    t = torch.tensor([0.5], dtype=torch.float64)

    def run(d):
        x = torch.tensor([[1.0, 0.5, 0.2], [0.3, -2.0, 1.0], [d, 0., 0.]], dtype=torch.float64)
        return torch.cat(R.score_head_forward(cfg, P, Ts, [R.FeaturedPoints(x, f, torch.zeros(3, dtype=torch.long))], qq, t))
    for d0 in (4.0, 5.0):
        a, b = run(d0 - 1e-7), run(d0 + 1e-7)
        assert (a - b).abs().max() < 1e-5
    # beyond the radius the point has no influence at all
    assert torch.equal(run(5.5), run(7.0))


def test_sampler_matches_manual_loop_and_appends_final_pose_twice():
    cfg, P, keys, q, Ts, time = _case(1, nT=3, n_scene=200, n_grasp=40)
    P32 = R.cast_params(P, torch.float32)
    keys32 = [R.FeaturedPoints(k.x.float(), k.f.float(), k.b) for k in keys]
    q32 = R.FeaturedPoints(q.x.float(), q.f.float(), q.b, q.w.float())
    g = torch.Generator().manual_seed(5)
    noise = torch.randn(3, 2, 3, 3, generator=g, dtype=torch.float64)
    traj = R.sample(cfg, P32, Ts, keys32, q32, [[1.0, 0.5]], [3], [0.04], temperatures=1.0, noise=noise)
    assert traj.shape == (5, 3, 7) and torch.equal(traj[-1], traj[-2]) and torch.equal(traj[0], Ts)
    assert torch.allclose(traj[..., :4].norm(dim=-1), torch.ones(5, 3, dtype=torch.float64), atol=1e-14)
    ts = R.t_schedule((1.0, 0.5), 3)
    assert abs(float(ts[0]) - 1.0) < 1e-15 and abs(float(ts[-1]) - 0.5) < 1e-15 and abs(float(ts[1]) - 0.5 ** 0.5) < 1e-15


def test_ebm_energy_is_se3_invariant():
    """compute_energy (score_head_ebm.py:122-174) is a scalar: invariant under a joint SE(3) move of scene and poses"""
    kw = synthetic.ebm_head_kwargs(2)
    cfgp = params.HeadConfig.from_kwargs(kw)
    cfg = R.config_from_kwargs(kw)
    P = params.init_params(cfgp, seed=2, randomize_all=True, dtype=torch.float64)
    keys = [R.FeaturedPoints(*k) for k in synthetic.make_key_clouds(cfgp, 400, dtype=torch.float64)]
    q = R.FeaturedPoints(*synthetic.make_query(cfgp, 80, dtype=torch.float64))
    Ts = synthetic.make_poses(5, near_object=True)
    t = torch.ones(5, dtype=torch.float64)
    E = R.compute_energy(cfg, P, Ts, keys, q, t)
    g, gt = _rand_q(3), torch.tensor([1., -2., 0.5], dtype=torch.float64)
    keys2 = [R.FeaturedPoints(R.quaternion_apply(g, k.x) + gt, R.transform_feature_quaternion(cfg.irreps, k.f, g[None])[0], k.b) for k in keys]
    Ts2 = torch.cat([R.quaternion_raw_multiply(g.expand(5, 4), Ts[:, :4]), R.quaternion_apply(g, Ts[:, 4:]) + gt], -1)
    assert (R.compute_energy(cfg, P, Ts2, keys2, q, t) - E).abs().max() < 1e-11
    assert (E > 0).all()


def test_segment_records_merge_to_the_joint_softmax():
    """The edge kernel emits, per run of same-destination edges inside a tile, the run's softmax-weighted mean value and the
    log-sum-exp of its logits; k_aggregate merges those records exactly as it would merge edges.  Check the identity the split
    relies on (graph_attention.py:247-262 joint softmax over all edges of a destination) for arbitrary run boundaries."""
    rng = np.random.default_rng(5)
    for trial in range(20):
        n = int(rng.integers(1, 60))
        logits = rng.normal(size=n) * 6.0
        vals = rng.normal(size=(n, 7))
        a = np.exp(logits - logits.max())
        direct = (a[:, None] * vals).sum(0) / a.sum()
        cuts = np.unique(np.concatenate([[0, n], rng.integers(0, n + 1, size=int(rng.integers(0, 6)))]))
        lse, mean = [], []
        for s, e in zip(cuts[:-1], cuts[1:]):
            m = logits[s:e].max()
            p = np.exp(logits[s:e] - m)
            lse.append(m + np.log(p.sum()))
            mean.append((p[:, None] * vals[s:e]).sum(0) / p.sum())
        lse, mean = np.array(lse), np.array(mean)
        w = np.exp(lse - lse.max())
        merged = (w[:, None] * mean).sum(0) / w.sum()
        assert np.allclose(merged, direct, rtol=1e-12, atol=1e-12)


def test_point_attention_multiplies_after_the_softmax():
    """use_src_point_attn (gnn_block.py:190-194, graph_attention.py:257-258): alpha = softmax(.) * w_src, NOT renormalised:
    unit weights change nothing, a common factor c scales the aggregated attention (hence the proj input) by c, and a zero
    weight removes a key point's contribution without changing the other edges' softmax weights."""
    from diffusion_edf_amd import params, synthetic
    kw = synthetic.score_head_kwargs(1, radii=(None,))
    kw['key_tensor_field_kwargs']['use_src_point_attn'] = True
    kw0 = synthetic.score_head_kwargs(1, radii=(None,))
    cfg, cfg0 = R.config_from_kwargs(kw), R.config_from_kwargs(kw0)
    assert cfg.use_src_point_attn and not cfg0.use_src_point_attn
    P = R.cast_params(params.init_params(params.HeadConfig.from_kwargs(kw), seed=2, randomize_all=True), torch.float64)
    g = torch.Generator().manual_seed(0)
    nK, nQ = 7, 5
    x, f = torch.randn(nK, 3, generator=g, dtype=torch.float64) * 5, torch.randn(nK, 160, generator=g, dtype=torch.float64)
    qx = torch.randn(nQ, 3, generator=g, dtype=torch.float64) * 5
    temb = [torch.randn(nQ, 64, generator=g, dtype=torch.float64)]
    b = torch.zeros(nK, dtype=torch.long)

    def attn(c, w):
        d = R.Debug()
        R.key_tensor_field(c, P, qx, [R.FeaturedPoints(x, f, b, w)], temb, d)
        return d['attn']
    base = attn(cfg0, None)
    assert torch.allclose(attn(cfg, torch.ones(nK, dtype=torch.float64)), base, atol=1e-14)
    assert torch.allclose(attn(cfg, torch.full((nK,), 0.25, dtype=torch.float64)), 0.25 * base, atol=1e-14)
    w = torch.ones(nK, dtype=torch.float64); w[3] = 0.0
    d = R.Debug()
    R.key_tensor_field(cfg0, P, qx, [R.FeaturedPoints(x, f, b, None)], temb, d)
    # plain attention minus key point 3's softmax-weighted share
    log_alpha, value, es, ed = d['log_alpha'], d['value'], d['edge_src'], d['edge_dst']
    Z = torch.zeros(nQ, log_alpha.shape[1], dtype=torch.float64).index_add_(0, ed, log_alpha.exp())
    share = (value * (log_alpha.exp() / Z[ed]).unsqueeze(-1))[es == 3]
    expect = base.clone()
    expect.index_add_(0, ed[es == 3], -R.heads2vec(share, [(m // cfg.num_heads, l) for m, l in cfg.irreps]))
    assert torch.allclose(attn(cfg, w), expect, atol=1e-12)
    with pytest.raises(AssertionError):
        attn(cfg, None)


def test_agent_sample_restatement_is_consistent():
    """oracle.agent_sample (agent.py:98-186): one stage == sample; two stages chain through the final poses and concatenate;
    the critic reorders the poses of every time step by ascending energy of the last step"""
    from diffusion_edf_amd import params, synthetic
    kw = synthetic.score_head_kwargs(1, radii=(4., None))
    cfg = params.HeadConfig.from_kwargs(kw)
    P = params.init_params(cfg, seed=3, randomize_all=True)
    keys = synthetic.make_key_clouds(cfg, 200, seed=0)
    query = synthetic.make_query(cfg, 30, seed=0)
    ok = [R.FeaturedPoints(k.x, k.f, k.b) for k in keys]
    oq = R.FeaturedPoints(query.x, query.f, query.b, query.w)
    ocfg = R.config_from_kwargs(kw)
    Ts = synthetic.make_poses(4, seed=1, near_object=True)
    noise = [torch.randn(2, 2, 4, 3, dtype=torch.float64, generator=torch.Generator().manual_seed(i)) for i in (1, 2)]
    m = (ocfg, P, ok, oq)
    one, e = R.agent_sample([m], None, Ts, [[2]], [[0.03]], [1.0], [[[1.0, 0.5]]], noise_list=noise[:1])
    assert e is None and torch.equal(one, R.sample(ocfg, P, Ts, ok, oq, [[1.0, 0.5]], [2], [0.03], 1.0, True, 1.0, 0.5, noise=noise[0]))
    two, _ = R.agent_sample([m, m], None, Ts, [[2], [2]], [[0.03], [0.02]], [1.0, 0.5], [[[1.0, 0.5]], [[0.5, 0.2]]], noise_list=noise)
    assert two.shape == (8, 4, 7) and torch.equal(two[:4], one) and torch.equal(two[4], one[-1])
    kwe = synthetic.ebm_head_kwargs(1, radii=(4., 6.))
    cfge = params.HeadConfig.from_kwargs(kwe)
    Pe = params.init_params(cfge, seed=5, randomize_all=True)
    ke = synthetic.make_key_clouds(cfge, 200, seed=0)
    crit = (R.config_from_kwargs(kwe), Pe, [R.FeaturedPoints(k.x, k.f, k.b) for k in ke], oq)
    ranked, es = R.agent_sample([m], crit, Ts, [[2]], [[0.03]], [1.0], [[[1.0, 0.5]]], noise_list=noise[:1])
    assert bool((es[1:] >= es[:-1]).all())
    e_direct = R.compute_energy(crit[0], Pe, one[-1].float(), crit[2], oq, torch.ones(4))
    order = torch.argsort(e_direct)
    assert torch.equal(ranked, one[:, order]) and torch.allclose(es, e_direct[order])
