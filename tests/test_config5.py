"""BASELINE config 5 as ONE model at its own size: MultiscaleScoreModel at lmax 3 (64x0e+32x1e+16x2e+8x3e, SH up to 3e) assembled from
YAML-shaped blocks (reference multiscale_score_model.py:27-112), the 16 384-point synthetic scene -> UnetFeatureExtractor -> key clouds
3277 / 656 / 132 / 27, a 1 024-point grasp cloud -> KeypointExtractor -> query EDF, 1 000 poses through ScoreModelBase.sample
(score_model_base.py:110-204) with the radial table in its default (automatic) mode -- what `bench.py --config5` times.

The fp64 chain oracle (U.unet_forward -> U.keypoint_extractor_forward -> R.score_head_forward -> R.langevin_step) anchors a pose subset at 1e-4
of the displacement scale (poses are independent units); the fp32 evaluation of the same chain gives the floor beside it.  The scene has ~90-190
edges per (pose, query) node (94 with poses uniform in the workspace), above the old default workspace of 96 per node: the
workspace is sized from the scene's own density (dedf_set_key_clouds) and must not overflow."""
import numpy as np
import pytest
import torch

from diffusion_edf_amd import synthetic
from oracle import restatement as R
from oracle import unet_oracle as U

IRREPS3 = [(64, 0), (32, 1), (16, 2), (8, 3)]
SH3 = [(1, 0), (1, 1), (1, 2), (1, 3)]


def test_config5_model_kwargs_are_in_the_reference_schema():
    """the YAML-shaped block parses into the three sub-model blocks multiscale_score_model.py:27-38 takes, without the injected keys"""
    kw = synthetic.config5_model_kwargs()
    assert set(kw) == {"score_head_kwargs", "key_kwargs", "query_model", "query_kwargs"}
    tf = kw["score_head_kwargs"]["key_tensor_field_kwargs"]
    assert "irreps_input" not in tf and "use_src_point_attn" not in tf and "irreps_query_edf" not in kw["score_head_kwargs"]
    assert tf["irreps_output"] == "64x0e+32x1e+16x2e+8x3e" and tf["irreps_sh"] == "1x0e+1x1e+1x2e+1x3e"
    fe = kw["key_kwargs"]["feature_extractor_kwargs"]
    assert fe["irreps_output"] == "64x0e+32x1e+16x2e+8x3e" and fe["irreps_emb"][0] == "32x0e+16x1e+8x2e+4x3e" and fe["pool_ratio"] == [0.2] * 4
    assert kw["query_kwargs"]["tensor_field_kwargs"]["irreps_output"] == "64x0e+32x1e+16x2e+8x3e"


def _unet_cfg(module):
    from diffusion_edf_amd.so3 import parse_irreps
    kw = module._ctor
    return U.UnetConfig(irreps_input=parse_irreps(kw["irreps_input"]), irreps_output=parse_irreps(kw["irreps_output"]),
                        irreps_emb=[parse_irreps(i) for i in kw["irreps_emb"]], fc_neurons=[list(f) for f in kw["fc_neurons"]],
                        n_layers=list(kw["n_layers"]), pool_ratio=list(kw["pool_ratio"]), radius=list(module.radius),
                        n_layers_midstream=kw["n_layers_midstream"], irreps_sh=SH3)


def _field_cfg(radii):
    return R.Config(irreps=IRREPS3, irreps_sh=SH3, num_heads=4, fc_neurons=[64, 32, 32], length_emb_dim=64, r_cluster_multiscale=list(radii),
                    r_mincut_nonscalar_sh=0.01 * radii[0], length_enc_max_r=None, time_emb_mlp=[256, 128, 64], max_time=1.0, time_enc_n=10000.0,
                    lin_mult=1.0, ang_mult=1.0, edge_time_encoding=False)


def build_config5(dev, n_scene=16384, n_grasp=1024, seed=0):
    """the model (seeded random-init weights of every sub-model, randomised so that no LayerNorm / bias sits at its init value) and its clouds"""
    from diffusion_edf_amd import agent as A, params
    from diffusion_edf_amd.gnn_data import FeaturedPoints
    from test_lmax3 import _randomized
    kw = synthetic.config5_model_kwargs()
    m = A.MultiscaleScoreModel(**kw, deterministic=True)
    _randomized(m.key_model, seed=1); _randomized(m.query_model, seed=2)
    hk = synthetic.score_head_kwargs(3)
    Ph = params.init_params(params.HeadConfig.from_kwargs(hk), seed=3, randomize_all=True)
    m.score_head.load_state_dict(Ph)
    m.to(dev).eval()
    scene = torch.from_numpy(synthetic.make_scene(n_scene, seed=seed).astype(np.float32))
    grasp = torch.from_numpy(synthetic.make_grasp(n_grasp, seed=seed).astype(np.float32))
    g = torch.Generator().manual_seed(0)
    fs, fg = torch.rand(len(scene), 3, generator=g), torch.rand(len(grasp), 3, generator=g)
    fp = lambda x, f: FeaturedPoints(x=x.to(dev), f=f.to(dev), b=torch.zeros(len(x), dtype=torch.long, device=dev), w=None)
    return m, kw, hk, Ph, (scene, fs), (grasp, fg), fp


@pytest.mark.gpu
def test_config5_chain_scene_to_denoised_poses_against_the_oracle():
    from diffusion_edf_amd.score_model_base import ScoreModelBase
    dev = torch.device("cuda:0")
    m, kw, hk, Ph, (scene, fs), (grasp, fg), fp = build_config5(dev)
    key = m.get_key_pcd_multiscale(fp(scene, fs))
    query = m.get_query_pcd(fp(grasp, fg))
    assert [len(k.x) for k in key] == [3277, 656, 132, 27] and len(query.x) == 103 and key[0].f.shape[1] == 296
    nT, t, dt = 1000, 0.5, 0.04
    Ts = synthetic.make_poses(nT, seed=1).to(dev)
    head = m.score_head
    out = m.sample(Ts, key, query, [[t, t]], [2], [dt], temperatures=0.0)
    st = head.stats()
    assert st["n_dst"] == nT * 103 and not st["overflow"] and not st["nonfinite"], st
    assert not any(st["rtab_fallback"]) and 0.0 < max(st["rtab_err"][:3]) < 1e-5, st          # the table was on (103 000 nodes) and its guard ran
    deg = st["n_edges_total"] / st["n_dst"]
    print(f"config 5: {st['n_edges_total']} edges per step = {deg:.1f} per (pose, query) node; per scale {st['n_edges']}")
    assert deg > 40.0          # (C2: 23; this scene is 4x as dense)
    # ---- the chain oracle, stage by stage, fp64 ----
    Pk = R.cast_params({k: v.cpu() for k, v in m.key_model.state_dict().items()}, torch.float64)
    Pq = R.cast_params({k: v.cpu() for k, v in m.query_model.state_dict().items()}, torch.float64)
    # (the key model IS the lmax-3 UNet of tests/test_lmax3.py with the same seeded weights, scene and features: one fp64 pass serves both tests)
    from test_lmax3 import unet_lmax3_reference
    rf = unet_lmax3_reference(16384)
    assert torch.equal(rf["x"], scene) and torch.equal(rf["f"], fs) and set(rf["sd"]) == set(Pk)
    assert all(torch.equal(rf["sd"][k].double(), Pk[k]) for k in Pk)
    key_ref = rf["ref"]
    radii = kw["query_kwargs"]["tensor_field_kwargs"]["r_cluster_multiscale"]
    xq, fq, wq = U.keypoint_extractor_forward(_unet_cfg(m.query_model.feature_extractor), _field_cfg(radii), Pq, grasp, fg.double(), 0.1, bbox=None)
    assert [len(k.x) for k in key] == [len(k[0]) for k in key_ref] and torch.equal(query.x.cpu(), xq)
    for k, (xr, fr) in zip(key, key_ref):
        assert torch.equal(k.x.cpu(), xr)
        print(f"TOLPROBE config5 keys: {float((k.f.cpu().double() - fr).abs().max()) / float(fr.abs().max()):.2e}")
        assert float((k.f.cpu().double() - fr).abs().max()) < 5e-5 * float(fr.abs().max())
    print(f"TOLPROBE config5 query: {float((query.f.cpu().double() - fq).abs().max()) / float(fq.abs().max()):.2e} {float((query.w.cpu().double() - wq).abs().max()):.2e}")
    assert float((query.f.cpu().double() - fq).abs().max()) < 5e-5 * float(fq.abs().max()) and float((query.w.cpu().double() - wq).abs().max()) < 5e-5
    sel = torch.tensor([0, 333, 777, 999])
    rcfg = R.config_from_kwargs(hk)
    kd = [R.FeaturedPoints(x=xr.double(), f=fr, b=torch.zeros(len(xr), dtype=torch.long), w=None) for xr, fr in key_ref]
    qd = R.FeaturedPoints(x=xq.double(), f=fq, b=torch.zeros(len(xq), dtype=torch.long), w=wq)
    Tsel = Ts[sel.to(dev)].cpu()
    tt = torch.full((len(sel),), t, dtype=torch.float64)
    P64 = R.cast_params(Ph, torch.float64)
    ang, lin = R.score_head_forward(rcfg, P64, Tsel, kd, qd, tt)
    z = torch.zeros(len(sel), 3, dtype=torch.float64)
    d_ref = R.langevin_step(rcfg, Tsel, ang, lin, t, dt, 0.0, 0.5, 0.5, z, z) - Tsel
    d_gpu = (out[1] - out[0])[sel.to(dev)].cpu()
    # the floor: the same chain's score head in fp32 on the fp64 chain's clouds (the 17-layer UNets' own fp32 floor is checked per scale above)
    a32, l32 = R.score_head_forward(rcfg, R.cast_params(Ph, torch.float32), Tsel.float(),
                                    [R.FeaturedPoints(x=k.x.float(), f=k.f.float(), b=k.b, w=None) for k in kd],
                                    R.FeaturedPoints(x=qd.x.float(), f=qd.f.float(), b=qd.b, w=qd.w.float()), tt.float())
    d_32 = R.langevin_step(rcfg, Tsel, a32.double(), l32.double(), t, dt, 0.0, 0.5, 0.5, z, z) - Tsel
    errs, floors = [], []
    for sl in (slice(0, 4), slice(4, 7)):          # rotation (quaternion) and translation parts, each against its own scale
        scale = float(d_ref[:, sl].abs().max())
        assert scale > 1e-4
        errs.append(float((d_gpu[:, sl] - d_ref[:, sl]).abs().max()) / scale)
        floors.append(float((d_32[:, sl] - d_ref[:, sl]).abs().max()) / scale)
    print(f"config 5 chain, one denoising step of a pose subset: HIP path {max(errs):.2e} of the displacement scale; fp32 restatement of the score head "
          f"on the same clouds (the floor) {max(floors):.2e}; bar 1e-4")
    assert max(errs) < 1e-4, (errs, floors)
    # the whole batch is finite and moved; the subset run alone reproduces its poses of the batch (pose independence ties the 1 000-pose run
    # to the oracle-checked values)
    assert torch.isfinite(out).all()
    head.set_radial_table("always")
    out_s = m.sample(Ts[sel.to(dev)], key, query, [[t, t]], [2], [dt], temperatures=0.0)
    moved = float((out[-1] - out[0]).abs().max())
    assert float((out_s[-1] - out[-1][sel.to(dev)]).abs().max()) < 2e-5 * moved


@pytest.mark.gpu
def test_config5_model_in_half_precision_mode():
    """`model.half()` (reference agent.py:50-51) on the WHOLE config-5 model: UNet layers, KeypointExtractor fields, lmax-3 score head -- every GEMM
    one fp16 MFMA product.  Scores within 5e-3 of the score scale of the full-precision run of the same model (whose parity the test above
    anchors), on a 4 096-point scene; key / query features per scale within 5e-3."""
    dev = torch.device("cuda:0")
    m, kw, hk, Ph, (scene, fs), (grasp, fg), fp = build_config5(dev, n_scene=4096)
    key = m.get_key_pcd_multiscale(fp(scene, fs))
    query = m.get_query_pcd(fp(grasp, fg))
    Ts = synthetic.make_poses(64, seed=1).to(dev)
    time = torch.linspace(0.1, 0.9, 64, device=dev)
    ang, lin = m.score_head(Ts.float(), key, query, time)
    m.score_head.half(); m.key_model.half(); m.query_model.half()
    key_h = m.get_key_pcd_multiscale(fp(scene, fs))
    query_h = m.get_query_pcd(fp(grasp, fg))
    for a, b in zip(key, key_h):
        assert torch.equal(a.x, b.x) and float((a.f - b.f).abs().max()) < 5e-3 * float(a.f.abs().max())
    assert torch.equal(query.x, query_h.x) and float((query.f - query_h.f).abs().max()) < 5e-3 * float(query.f.abs().max())
    assert float((query.w - query_h.w).abs().max()) < 5e-3
    # the head alone on the full-precision clouds, then the whole half-precision chain
    ang_h, lin_h = m.score_head(Ts.float(), key, query, time)
    scale = float(max(ang.abs().max(), lin.abs().max()))
    e_head = max(float((ang_h - ang).abs().max()), float((lin_h - lin).abs().max())) / scale
    ang_c, lin_c = m.score_head(Ts.float(), key_h, query_h, time)
    e_chain = max(float((ang_c - ang).abs().max()), float((lin_c - lin).abs().max())) / scale
    print(f"config 5 in half-precision mode: score head alone {e_head:.2e}, whole chain {e_chain:.2e} of the score scale")
    assert e_head < 5e-3 and e_chain < 2e-2, (e_head, e_chain)
    assert not m.score_head.stats()["nonfinite"]
