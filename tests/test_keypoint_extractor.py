"""KeypointExtractor (keypoint_extractor.py:50-197): the place tasks' query model = UNet + FPS key points + two context-free
MultiscaleTensorFields + the weight head.  CPU: schema / construction / oracle sanity.  GPU: dedf_field and the whole extractor against the fp64
restatement."""
import math

import numpy as np
import pytest
import torch

from diffusion_edf_amd import synthetic
from oracle import restatement as R
from oracle import unet_oracle as U
from test_unet import _oracle_cfg, _randomized, _unet_kwargs

WIDE = "64x0e+32x1e+16x2e"


_query_kwargs = synthetic.keypoint_extractor_kwargs


def _field_cfg(radii):
    irr = R.parse_irreps(WIDE)
    return R.Config(irreps=irr, irreps_sh=R.parse_irreps("1x0e+1x1e+1x2e"), num_heads=4, fc_neurons=[64, 32, 32], length_emb_dim=64,
                    r_cluster_multiscale=list(radii), r_mincut_nonscalar_sh=0.01 * radii[0], length_enc_max_r=None, time_emb_mlp=[256, 128, 64],
                    max_time=1.0, time_enc_n=10000.0, lin_mult=1.0, ang_mult=1.0, edge_time_encoding=False)


def test_keypoint_extractor_schema():
    from diffusion_edf_amd.keypoint_extractor import KeypointExtractor, MultiscaleTensorField, StaticKeypointModel
    m = KeypointExtractor(**_query_kwargs(), deterministic=True)
    sd = m.state_dict()
    names = set(sd)
    for k in ("feature_extractor.input_emb.tp.weight", "feature_extractor.mid_block.0.gnn.ga.alpha_dot",
              "tensor_field.graph_parsers.3.length_enc.param_module.mean", "tensor_field.edge_scalars_pre_linears.0.0.weight",
              "tensor_field.gnn_block_init.prenorm_src.affine_weight", "tensor_field.gnn_block_init.linear_src.bias.0",
              "tensor_field.gnn_block_init.ga.sep_act.dtp_rad.net.0.weight", "tensor_field.gnn_block_init.post_norm.affine_bias",
              "tensor_field.gnn_block_init.ffn.fctp_2.tp.weight", "weight_field.gnn_block_init.skip_2.skip.tp.weight",
              "weight_field.gnn_block_init.skip_2.skip.bias.0", "weight_post.0.weight", "weight_post.2.bias"):
        assert k in names, k
    assert not any(k.startswith("tensor_field.gnn_block_init.skip_2") or "time_mlps" in k or "key_tensor_field" in k for k in names)
    assert sd["tensor_field.edge_scalars_pre_linears.2.0.weight"].shape == (64, 64)                  # no context: fc_neurons[0] = length_emb_dim
    assert sd["tensor_field.gnn_block_init.ga.sep_act.dtp_rad.net.0.weight"].shape == (32, 64)
    assert sd["tensor_field.gnn_block_init.ffn.fctp_2.tp.weight"].numel() == 192 * 64 + 96 * 32 + 48 * 16
    assert sd["weight_field.gnn_block_init.ffn.fctp_2.tp.weight"].numel() == 192 * 64                # ends in 64x0e
    assert sd["weight_field.gnn_block_init.skip_2.skip.tp.weight"].numel() == 64 * 64
    assert "weight_mult_logit" not in names
    m2 = KeypointExtractor(**dict(_query_kwargs(), weight_mult=2.0), deterministic=True)
    assert abs(float(torch.nn.functional.softplus(m2.state_dict()["weight_mult_logit"])) - 2.0) < 1e-6
    with pytest.raises(NotImplementedError):
        MultiscaleTensorField(irreps_input=WIDE, irreps_output=WIDE, irreps_sh="1x0e+1x1e+1x2e", num_heads=4, fc_neurons=[-1, 32, 32],
                              length_emb_dim=64, irreps_query=WIDE, r_cluster_multiscale=[5.0], edge_context_emb_dim=None)
    with pytest.raises(RuntimeError):                        # no CPU path
        from diffusion_edf_amd.gnn_data import FeaturedPoints
        m(FeaturedPoints(x=torch.zeros(50, 3), f=torch.zeros(50, 3), b=torch.zeros(50, dtype=torch.long), w=None))
    s = StaticKeypointModel([[0.0, 0.0, 1.0], [1.0, 0.0, 0.0]], WIDE)
    from diffusion_edf_amd.gnn_data import FeaturedPoints
    out = s(FeaturedPoints(x=torch.zeros(4, 3), f=torch.zeros(4, 3), b=torch.zeros(4, dtype=torch.long), w=None))
    assert out.x.shape == (2, 3) and out.f.shape == (2, 240) and out.w.shape == (2,) and bool(((out.w > 0) & (out.w < 1)).all())
    assert set(s.state_dict()) == {"keypoint_coords", "keypoint_features", "keypoint_weights"}


def _object_cloud(n, seed):
    """a grasped-object-sized cloud: the synthetic scene shrunk into the bbox region"""
    x = synthetic.make_scene(n, seed=seed).astype(np.float32)
    x = (x - x.mean(0)) * 0.5
    x[:, 2] += 14.0 - x[:, 2].min()
    return torch.from_numpy(x.astype(np.float32))


def test_keypoint_extractor_oracle_small():
    """weights in (0,1), key points inside the bbox, the weight field's skip_2 really is the LinearRS (zeroing it changes the weights), and the
    field at a point far from everything is the bias-only value (no edges: attention output 0 -> emb = proj bias)"""
    from diffusion_edf_amd.keypoint_extractor import KeypointExtractor
    radii = (5.0, 10.0, 20.0, 40.0)
    m = KeypointExtractor(**_query_kwargs(radii), deterministic=True)
    sd = _randomized(m, seed=2)
    P = R.cast_params(sd, torch.float64)
    x = _object_cloud(1200, seed=1)
    x[:, 2] -= 8.0                                                      # part of the cloud below the bbox floor
    f = torch.rand(len(x), 3, dtype=torch.float64, generator=torch.Generator().manual_seed(0))
    bbox = _query_kwargs()["keypoint_kwargs"]["bbox"]
    xq, feat, w = U.keypoint_extractor_forward(_oracle_cfg(m.feature_extractor), _field_cfg(radii), P, x, f, 0.1, bbox=bbox)
    n_in = int(((x >= torch.tensor(bbox)[:, 0]) & (x <= torch.tensor(bbox)[:, 1])).all(-1).sum())
    assert 0 < n_in < len(x) and len(xq) == math.ceil(0.1 * n_in) and float(xq[:, 2].min()) >= 8.0
    assert feat.shape == (len(xq), 240) and w.shape == (len(xq),) and bool(((w > 0) & (w < 1)).all()) and bool(torch.isfinite(feat).all())
    P0 = dict(P)
    P0["weight_field.gnn_block_init.skip_2.skip.tp.weight"] = P0["weight_field.gnn_block_init.skip_2.skip.tp.weight"] * 0
    _, feat0, w0 = U.keypoint_extractor_forward(_oracle_cfg(m.feature_extractor), _field_cfg(radii), P0, x, f, 0.1, bbox=bbox)
    assert torch.equal(feat0, feat) and float((w0 - w).abs().max()) > 1e-4


@pytest.mark.gpu
def test_dedf_field_matches_the_oracle_and_handles_isolated_points():
    """MultiscaleTensorField.forward on the HIP path (dedf_field) against the restatement, on random key clouds at four finite scales; the query
    points include some beyond every radius (no edges): their field is the bias-only value, as the reference's scatter gives"""
    from diffusion_edf_amd.gnn_data import FeaturedPoints
    from diffusion_edf_amd.keypoint_extractor import MultiscaleTensorField
    dev = torch.device("cuda:0")
    radii = (5.0, 10.0, 20.0, 40.0)
    g = torch.Generator().manual_seed(0)
    key = [FeaturedPoints(x=torch.rand(n, 3, generator=g) * 50.0, f=torch.randn(n, 240, generator=g), b=torch.zeros(n, dtype=torch.long), w=None)
           for n in (900, 300, 90, 30)]
    xq = torch.cat([torch.rand(200, 3, generator=g) * 50.0, torch.tensor([[500.0, 0.0, 0.0], [0.0, -400.0, 90.0]])])
    for scalar_out in (False, True):
        m = MultiscaleTensorField(irreps_input=WIDE, irreps_output="64x0e" if scalar_out else WIDE, irreps_sh="1x0e+1x1e+1x2e", num_heads=4,
                                  fc_neurons=[-1, 32, 32], length_emb_dim=64, irreps_query=None, r_cluster_multiscale=list(radii),
                                  edge_context_emb_dim=None)
        sd = _randomized(m, seed=7)
        P = R.cast_params({"tf." + k: v for k, v in sd.items()}, torch.float64)
        kd = [R.FeaturedPoints(x=p.x.double(), f=p.f.double(), b=p.b, w=None) for p in key]
        ref = R.key_tensor_field(_field_cfg(radii), P, xq.double(), kd, None, pre="tf", irreps_output=[(64, 0)] if scalar_out else None)
        m.to(dev)
        out = m(FeaturedPoints(x=xq.to(dev), f=torch.empty(len(xq), 3, device=dev), b=torch.zeros(len(xq), dtype=torch.long, device=dev), w=None),
                [FeaturedPoints(x=p.x.to(dev), f=p.f.to(dev), b=p.b.to(dev), w=None) for p in key])
        got = out.f.cpu().double()
        assert got.shape == ref.shape and torch.equal(out.x.cpu(), xq)
        err = float((got - ref).abs().max()) / float(ref.abs().max())
        assert err < 1e-4, (scalar_out, err)
        assert float((got[-2:] - ref[-2:]).abs().max()) < 1e-5 * float(ref.abs().max())          # isolated points
        assert float((ref[-1] - ref[-2]).abs().max()) < 1e-12                                      # ... all get the same bias-only value


@pytest.mark.gpu
@pytest.mark.parametrize("n_points,bbox", [(3500, True), (2000, False)])
def test_keypoint_extractor_matches_the_oracle(n_points, bbox):
    from diffusion_edf_amd.gnn_data import FeaturedPoints
    from diffusion_edf_amd.keypoint_extractor import KeypointExtractor
    dev = torch.device("cuda:0")
    radii = (5.0, 10.0, 20.0, 40.0)
    kw = _query_kwargs(radii) if bbox else _query_kwargs(radii, bbox=None)
    m = KeypointExtractor(**kw, deterministic=True)
    sd = _randomized(m, seed=9)
    x = _object_cloud(n_points, seed=3)
    if bbox:
        x[:, 2] -= 8.0
    f = torch.rand(n_points, 3, generator=torch.Generator().manual_seed(1))
    xr, fr, wr = U.keypoint_extractor_forward(_oracle_cfg(m.feature_extractor), _field_cfg(radii), R.cast_params(sd, torch.float64), x, f.double(),
                                              0.1, bbox=kw["keypoint_kwargs"]["bbox"])
    m.to(dev)
    out = m(FeaturedPoints(x=x.to(dev), f=f.to(dev), b=torch.zeros(n_points, dtype=torch.long, device=dev), w=None))
    assert torch.equal(out.x.cpu(), xr) and out.f.shape == fr.shape and out.w.shape == wr.shape
    got = out.f.cpu().double()
    off = 0
    for mul, l in R.parse_irreps(WIDE):
        d = mul * (2 * l + 1)
        err = float((got[:, off:off + d] - fr[:, off:off + d]).abs().max()) / float(fr[:, off:off + d].abs().max())
        print(f"TOLPROBE keypoint f l={l}: {err:.2e}")
        assert err < 5e-5, (l, err)
        off += d
    print(f"TOLPROBE keypoint w: {float((out.w.cpu().double() - wr).abs().max()):.2e}")
    assert float((out.w.cpu().double() - wr).abs().max()) < 5e-5
    assert float(wr.max() - wr.min()) > 1e-3                       # the weights are not a constant


@pytest.mark.gpu
def test_whole_place_model_from_clouds_to_scores():
    """MultiscaleScoreModel assembled from YAML-shaped blocks (UNet key model, KeypointExtractor query model, score head), every stage on the HIP
    path: scene cloud -> key_pcd_multiscale, grasp cloud -> query_pcd, (poses, time) -> (ang, lin), against the chained restatement, at the
    north-star bar of the score (1e-4; measured 1.2e-5 with the fp32 restatement of the same chain at 1.3e-5 -- round 2 held this to 2e-3)."""
    from diffusion_edf_amd import agent as A, params
    from diffusion_edf_amd.gnn_data import FeaturedPoints
    from test_agent import _model_yaml
    dev = torch.device("cuda:0")
    radii = (5.0, 10.0, 20.0, 40.0)
    hk = synthetic.score_head_kwargs(2)
    doc = _model_yaml(hk)["model_kwargs"]
    doc["query_model"], doc["query_kwargs"] = "KeypointExtractor", _query_kwargs(radii)
    m = A.MultiscaleScoreModel(**doc, deterministic=True)
    assert set(k.split(".")[0] for k in m.state_dict()) == {"key_model", "query_model", "score_head"}
    _randomized(m.key_model, seed=1); _randomized(m.query_model, seed=2)
    hcfg = params.HeadConfig.from_kwargs(hk)
    Ph = params.init_params(hcfg, seed=3, randomize_all=True)
    m.score_head.load_state_dict(Ph)
    m.to(dev).eval()
    scene = torch.from_numpy(synthetic.make_scene(5000, seed=5).astype(np.float32))
    grasp = _object_cloud(2500, seed=6)
    grasp[:, 2] -= 8.0
    g = torch.Generator().manual_seed(0)
    fs, fg = torch.rand(len(scene), 3, generator=g), torch.rand(len(grasp), 3, generator=g)
    fp = lambda x, f: FeaturedPoints(x=x.to(dev), f=f.to(dev), b=torch.zeros(len(x), dtype=torch.long, device=dev), w=None)
    key = m.get_key_pcd_multiscale(fp(scene, fs))
    query = m.get_query_pcd(fp(grasp, fg))
    # restatement, stage by stage
    Pk = R.cast_params({k: v.cpu() for k, v in m.key_model.state_dict().items()}, torch.float64)
    Pq = R.cast_params({k: v.cpu() for k, v in m.query_model.state_dict().items()}, torch.float64)
    key_ref = U.unet_forward(_oracle_cfg(m.key_model), Pk, scene, fs.double())
    xq, fq, wq = U.keypoint_extractor_forward(_oracle_cfg(m.query_model.feature_extractor), _field_cfg(radii), Pq, grasp, fg.double(), 0.1,
                                              bbox=doc["query_kwargs"]["keypoint_kwargs"]["bbox"])
    assert [len(k.x) for k in key] == [len(k[0]) for k in key_ref] and torch.equal(query.x.cpu(), xq)
    for k, (xr, fr) in zip(key, key_ref):
        print(f"TOLPROBE model keys: {float((k.f.cpu().double() - fr).abs().max()) / float(fr.abs().max()):.2e}")
        assert float((k.f.cpu().double() - fr).abs().max()) < 5e-5 * float(fr.abs().max())
    print(f"TOLPROBE model query: {float((query.f.cpu().double() - fq).abs().max()) / float(fq.abs().max()):.2e} {float((query.w.cpu().double() - wq).abs().max()):.2e}")
    assert float((query.f.cpu().double() - fq).abs().max()) < 5e-5 * float(fq.abs().max()) and float((query.w.cpu().double() - wq).abs().max()) < 5e-5
    nT = 24
    q = torch.randn(nT, 4, generator=g, dtype=torch.float64)
    Ts = torch.cat([q / q.norm(dim=-1, keepdim=True), torch.randn(nT, 3, generator=g, dtype=torch.float64) * 6.0 + torch.tensor([0.0, 0.0, 4.0])], -1)
    time = torch.rand(nT, generator=g, dtype=torch.float64) * 0.9 + 0.05
    ang, lin = m.score_head(Ts.float().to(dev), key, query, time.float().to(dev))
    rcfg = R.config_from_kwargs(hk)
    kd = [R.FeaturedPoints(x=xr.double(), f=fr, b=torch.zeros(len(xr), dtype=torch.long), w=None) for xr, fr in key_ref]
    qd = R.FeaturedPoints(x=xq.double(), f=fq, b=torch.zeros(len(xq), dtype=torch.long), w=wq)
    ang_r, lin_r = R.score_head_forward(rcfg, R.cast_params(Ph, torch.float64), Ts, kd, qd, time)
    # The error FLOOR of the chain: the same restatement evaluated in fp32 end to end (UNet 17 layers deep -> key clouds; UNet + two fields ->
    # query EDF; score head), against its own fp64 run: printed beside the HIP path's error as a diagnostic (the asserted bar is the stated 1e-4).
    f32 = torch.float32
    key_32 = U.unet_forward(_oracle_cfg(m.key_model), R.cast_params(Pk, f32), scene, fs)
    xq32, fq32, wq32 = U.keypoint_extractor_forward(_oracle_cfg(m.query_model.feature_extractor), _field_cfg(radii), R.cast_params(Pq, f32), grasp, fg, 0.1,
                                                    bbox=doc["query_kwargs"]["keypoint_kwargs"]["bbox"])
    floor = None
    if [len(k[0]) for k in key_32] == [len(k[0]) for k in key_ref] and torch.equal(xq32, xq):       # (same graphs: integer work on the same fp32 coordinates)
        k32 = [R.FeaturedPoints(x=xr, f=fr, b=torch.zeros(len(xr), dtype=torch.long), w=None) for xr, fr in key_32]
        q32 = R.FeaturedPoints(x=xq32, f=fq32, b=torch.zeros(len(xq32), dtype=torch.long), w=wq32)
        a32, l32 = R.score_head_forward(rcfg, R.cast_params(Ph, f32), Ts.float(), k32, q32, time.float())
        floor = max(float((a32.double() - ang_r).abs().max()) / float(ang_r.abs().max()), float((l32.double() - lin_r).abs().max()) / float(lin_r.abs().max()))
    errs = []
    for got, ref in ((ang, ang_r), (lin, lin_r)):
        err = float((got.cpu().double() - ref).abs().max()) / float(ref.abs().max())
        errs.append(err)
        assert err < 1e-4, err
    print(f"whole chain clouds -> scores: HIP path {max(errs):.2e} of the score scale; fp32 restatement (the floor) {floor if floor is None else format(floor, '.2e')}; bar 1e-4")


@pytest.mark.gpu
def test_whole_point_attentive_model_from_clouds_to_scores():
    """PointAttentiveScoreModel (the sapien *_lowres configs) assembled from YAML-shaped blocks: the key model is ONE KeypointExtractor whose
    weighted key points form the single, all-pairs key cloud (point_attentive_score_model.py:34-37,106-107); the query model a StaticKeypointModel"""
    from diffusion_edf_amd import agent as A, params
    from diffusion_edf_amd.gnn_data import FeaturedPoints
    from test_agent import _model_yaml
    dev = torch.device("cuda:0")
    radii = (5.0, 10.0, 20.0, 40.0)
    hk = synthetic.score_head_kwargs(2, radii=(None,))
    hk["key_tensor_field_kwargs"]["fc_neurons"] = [-1, 32, 32]                     # configs/sapien/place_lowres/score_model_configs.yaml:15
    doc = _model_yaml(hk)["model_kwargs"]
    doc["key_kwargs"] = dict(_query_kwargs(radii, bbox=None, pool_ratio=0.05), feature_extractor_name="UnetFeatureExtractor")
    m = A.PointAttentiveScoreModel(**doc, deterministic=True)
    _randomized(m.key_model, seed=4)
    hk_full = copy_with_point_attn(hk)
    hcfg = params.HeadConfig.from_kwargs(hk_full)
    Ph = params.init_params(hcfg, seed=5, randomize_all=True)
    m.score_head.load_state_dict(Ph)
    m.to(dev).eval()
    scene = torch.from_numpy(synthetic.make_scene(4000, seed=8).astype(np.float32))
    g = torch.Generator().manual_seed(0)
    fs = torch.rand(len(scene), 3, generator=g)
    pcd = FeaturedPoints(x=scene.to(dev), f=fs.to(dev), b=torch.zeros(len(scene), dtype=torch.long, device=dev), w=None)
    key = m.get_key_pcd_multiscale(pcd)
    query = m.get_query_pcd(pcd)
    assert len(key) == 1 and len(key[0].x) == 200 and key[0].w is not None and len(query.x) == 2
    Pk = R.cast_params({k: v.cpu() for k, v in m.key_model.state_dict().items()}, torch.float64)
    xk, fk, wk = U.keypoint_extractor_forward(_oracle_cfg(m.key_model.feature_extractor), _field_cfg(radii), Pk, scene, fs.double(), 0.05, bbox=None)
    print(f"TOLPROBE key w: {float((key[0].w.cpu().double() - wk).abs().max()):.2e}")
    assert torch.equal(key[0].x.cpu(), xk) and float((key[0].w.cpu().double() - wk).abs().max()) < 5e-5
    nT = 16
    q = torch.randn(nT, 4, generator=g, dtype=torch.float64)
    Ts = torch.cat([q / q.norm(dim=-1, keepdim=True), torch.randn(nT, 3, generator=g, dtype=torch.float64) * 6.0 + torch.tensor([0.0, 0.0, 4.0])], -1)
    time = torch.rand(nT, generator=g, dtype=torch.float64) * 0.9 + 0.05
    ang, lin = m.score_head(Ts.float().to(dev), key, query, time.float().to(dev))
    rcfg = R.config_from_kwargs(hk_full)
    kd = [R.FeaturedPoints(x=xk.double(), f=fk, b=torch.zeros(len(xk), dtype=torch.long), w=wk)]
    qd = R.FeaturedPoints(x=query.x.cpu().double(), f=query.f.cpu().double(), b=torch.zeros(len(query.x), dtype=torch.long), w=query.w.cpu().double())
    ang_r, lin_r = R.score_head_forward(rcfg, R.cast_params(Ph, torch.float64), Ts, kd, qd, time)
    for got, ref in ((ang, ang_r), (lin, lin_r)):
        err = float((got.cpu().double() - ref).abs().max()) / float(ref.abs().max())
        assert err < 1e-4, err


def copy_with_point_attn(hk):
    import copy
    out = copy.deepcopy(hk)
    out["key_tensor_field_kwargs"]["use_src_point_attn"] = True
    return out
