"""Host-side callers of the path (diffusion_edf_amd/agent.py): model assembly from YAML + checkpoint as reference
trainer.py:35-70,124-147 / agent.py:20-64 do it, and the argument contract of DiffusionEdfAgent.sample (agent.py:98-128).
No GPU: heads are only constructed, never evaluated (the product has no CPU compute path)."""
import copy
import glob
import os

import pytest
import torch
import yaml

from diffusion_edf_amd import agent as A
from diffusion_edf_amd import params, synthetic
from diffusion_edf_amd.gnn_data import FeaturedPoints
from test_unet import _unet_kwargs

CFG_ROOT = "/root/reference/configs"
DIRS = sorted(os.path.dirname(f) for f in glob.glob(os.path.join(CFG_ROOT, "*", "*", "score_model_configs.yaml")))


def _model_yaml(kw):
    """score_model_configs.yaml in the reference's schema (configs/panda_mug/pick_lowres/score_model_configs.yaml:1-80) from
    synthetic kwargs: the keys multiscale_score_model.py:79-85 injects are absent from the file"""
    sh = copy.deepcopy(kw)
    irr = sh.pop('irreps_query_edf')
    tf = sh['key_tensor_field_kwargs']
    for k in ('irreps_input', 'use_src_point_attn'):
        tf.pop(k, None)
    return dict(model_name='MultiscaleScoreModel',
                model_kwargs=dict(score_head_kwargs=sh,
                                  key_kwargs=dict(feature_extractor_name='UnetFeatureExtractor', feature_extractor_kwargs=_unet_kwargs("panda_lowres")),
                                  query_model='StaticKeypointModel',
                                  query_kwargs=dict(keypoint_coords=[[0.5, 0.5, 10.5], [-0.5, -0.5, 10.5]], irreps_output=irr)))


def _write_config_dir(d, kw, schedules=((1.0, 0.15),)):
    os.makedirs(d, exist_ok=True)
    yaml.safe_dump(dict(model_config_file='score_model_configs.yaml', device='cuda:0',
                        diffusion_configs=dict(time_schedules=[list(s) for s in schedules], t_augment=0.01)), open(os.path.join(d, 'train_configs.yaml'), 'w'))
    yaml.safe_dump(dict(task_type='pick'), open(os.path.join(d, 'task_configs.yaml'), 'w'))
    yaml.safe_dump(_model_yaml(kw), open(os.path.join(d, 'score_model_configs.yaml'), 'w'))


@pytest.mark.skipif(not DIRS, reason="reference tree not present")
@pytest.mark.parametrize("d", DIRS, ids=[os.path.relpath(d, CFG_ROOT) for d in DIRS])
def test_get_models_on_shipped_config_dirs(d):
    doc = yaml.safe_load(open(os.path.join(d, "score_model_configs.yaml")))
    m = A.get_models(d, "train_configs.yaml", "task_configs.yaml", None, "cpu", n_warmups=0)
    train = yaml.safe_load(open(os.path.join(d, "train_configs.yaml")))
    assert m.diffusion_schedules == train["diffusion_configs"]["time_schedules"]
    sh = doc["model_kwargs"]["score_head_kwargs"]
    assert m.lin_mult == float(sh["lin_mult"]) and m.ang_mult == float(sh["ang_mult"])
    assert isinstance(m.score_head, A.EbmScoreModelHead) == bool(sh.get("ebm", False))
    assert isinstance(m, A.PointAttentiveScoreModel) == (doc["model_name"] == "PointAttentiveScoreModel") == m.score_head.cfg.use_src_point_attn
    assert m.score_head.cfg.radii == [None if r is None else float(r) for r in sh["key_tensor_field_kwargs"]["r_cluster_multiscale"]]
    assert not m.training
    # the feature extractors are built from the YAML blocks like the reference builds them (multiscale_score_model.py:40-62,
    # point_attentive_score_model.py:34-50), under the reference's state-dict prefixes; they run on the GPU only
    from diffusion_edf_amd.gnn_data import FeaturedPoints
    from diffusion_edf_amd.keypoint_extractor import KeypointExtractor
    from diffusion_edf_amd.unet import ForwardOnlyFeatureExtractor, UnetFeatureExtractor
    kk = doc["model_kwargs"]["key_kwargs"]
    want = KeypointExtractor if doc["model_name"] == "PointAttentiveScoreModel" else \
        {"UnetFeatureExtractor": UnetFeatureExtractor, "ForwardOnlyFeatureExtractor": ForwardOnlyFeatureExtractor}[kk["feature_extractor_name"]]
    assert type(m.key_model) is want
    names = set(m.state_dict())
    unet_prefix = "key_model.feature_extractor." if want is KeypointExtractor else "key_model."
    assert unet_prefix + "input_emb.tp.weight" in names and unet_prefix + "down_blocks.0.pool_layer.gnn.ga.alpha_dot" in names
    assert (unet_prefix + "mid_block.0.radial.mean" in names) == (kk["feature_extractor_name"] == "UnetFeatureExtractor")
    if want is KeypointExtractor:
        assert "key_model.weight_field.gnn_block_init.skip_2.skip.tp.weight" in names and "key_model.weight_post.2.weight" in names
    cloud = FeaturedPoints(x=torch.zeros(5, 3), f=torch.zeros(5, 3), b=torch.zeros(5, dtype=torch.long))
    with pytest.raises(RuntimeError):                  # no CPU path, and no silent stand-in
        m.get_key_pcd_multiscale(cloud)
    if doc["model_kwargs"]["query_model"] == "StaticKeypointModel":       # parameters only: part of the model (keypoint_extractor.py:22-47)
        qk = doc["model_kwargs"]["query_kwargs"]
        q = m.get_query_pcd(cloud)
        nK = len(qk["keypoint_coords"])
        assert q.x.shape == (nK, 3) and q.f.shape == (nK, m.score_head.cfg.dim) and q.w.shape == (nK,) and q.b.shape == (nK,)
        assert torch.equal(q.x, torch.tensor(qk["keypoint_coords"])) and bool(((q.w > 0) & (q.w < 1)).all())
        assert {k for k in m.state_dict() if k.startswith("query_model.")} == {
            "query_model.keypoint_coords", "query_model.keypoint_features", "query_model.keypoint_weights"}
    else:
        assert type(m.query_model) is KeypointExtractor and "query_model.tensor_field.gnn_block_init.ga.alpha_dot" in names
        with pytest.raises(RuntimeError):
            m.get_query_pcd(cloud)


def _reference_buffers(model_sd, cfg):
    """keys of persistent BUFFERS a reference checkpoint carries next to the parameters (constants, no trained state): e3nn's
    `tp.output_mask` for every TensorProduct, `cutoff_eps` of every graph parser (graph_parser.py:37), the Wigner J of every
    SliceAndTransform (wigner.py:215), constants of e3nn's generated code"""
    extra = {}
    for k in model_sd:
        if k.endswith(".tp.weight"):
            extra[k[:-len("weight")] + "output_mask"] = torch.ones(7)
            extra[k[:-len("weight")] + "_compiled_main_left_right._w3j_1_1_0"] = torch.ones(3, 3, 1)
    for n in range(cfg.n_scales):
        extra[f"score_head.key_tensor_field.graph_parsers.{n}.cutoff_eps"] = torch.tensor(1e-12)
    for i, (_, l) in enumerate(cfg.irreps):
        extra[f"score_head.transform_irreps.transforms.{i}.J"] = torch.eye(2 * l + 1)
    return extra


def test_checkpoint_load_like_the_reference_agent(tmp_path):
    kw = synthetic.score_head_kwargs(2)
    d = str(tmp_path / "pick_lowres")
    _write_config_dir(d, kw, schedules=((1.0, 0.15), (0.15, 0.01)))
    cfg = params.HeadConfig.from_kwargs(kw)
    P = params.init_params(cfg, seed=11, randomize_all=True)
    # the full reference model: key model (built here from the YAML block, so its names are this build's schema), static key points, head
    ref = A.get_models(d, "train_configs.yaml", "task_configs.yaml", None, "cpu", n_warmups=0)
    g = torch.Generator().manual_seed(5)
    sd = {k: v + 0.01 * torch.randn(v.shape, generator=g) if v.is_floating_point() else v.clone()
          for k, v in ref.state_dict().items() if k.startswith("key_model.")}
    sd.update({"score_head." + k: v for k, v in P.items()})
    sd["query_model.keypoint_coords"] = torch.tensor([[0.5, 0.5, 10.5], [-0.5, -0.5, 10.5]])
    sd["query_model.keypoint_features"] = torch.randn(2, cfg.dim)
    sd["query_model.keypoint_weights"] = torch.tensor([0.3, -1.2])
    # ... and the persistent buffers a genuine reference checkpoint always carries: tolerated, also under strict_load=True
    sd_ref = dict(sd, **_reference_buffers(sd, cfg))
    ck = str(tmp_path / "Pick_LowRes_300.pt")
    torch.save(dict(score_model_state_dict=sd_ref, epoch=300, steps=12345), ck)
    for strict in (False, True):
        m = A.get_models(d, "train_configs.yaml", "task_configs.yaml", ck, "cpu", n_warmups=0, strict_load=strict)   # default False: agent.py:28
        got = m.score_head.state_dict()
        assert set(got) == set(P)
        for k, v in P.items():
            assert torch.equal(got[k], v), k
        for k, v in m.state_dict().items():
            if k.startswith("key_model."):
                assert torch.equal(v, sd[k]), k                                                # the extractor really took the checkpoint's values
    q = m.get_query_pcd(synthetic.make_query(cfg, 0, seed=0, static_keypoints=True))        # StaticKeypointModel: weights from the checkpoint
    assert torch.equal(q.f, sd["query_model.keypoint_features"]) and torch.allclose(q.w, torch.sigmoid(torch.tensor([0.3, -1.2])))
    assert m.diffusion_schedules == [[1.0, 0.15], [0.15, 0.01]]
    # an INJECTED key extractor is the caller's: its keys are not this build's to judge under strict_load=False
    inj = {k: v for k, v in sd.items() if not k.startswith("key_model.")}
    inj["key_model.blocks.0.weight"] = torch.zeros(3)
    torch.save(dict(score_model_state_dict=inj, epoch=1, steps=1), ck)
    A.get_models(d, "train_configs.yaml", "task_configs.yaml", ck, "cpu", n_warmups=0, key_extractor=lambda pcd: None)
    with pytest.raises(RuntimeError, match="do not match the schema"):          # ... but a key model built here must find its parameters
        A.get_models(d, "train_configs.yaml", "task_configs.yaml", ck, "cpu", n_warmups=0)
    with pytest.raises(RuntimeError, match="Unexpected key"):
        A.get_models(d, "train_configs.yaml", "task_configs.yaml", ck, "cpu", n_warmups=0, strict_load=True, key_extractor=lambda pcd: None)
    # a renamed / missing extractor parameter is as fatal as a head parameter: that module would silently run on its seeded init
    k0 = next(k for k in sd if k.startswith("key_model.") and k.endswith("ga.alpha_dot"))
    ren = {(k0 + "_renamed" if k == k0 else k): v for k, v in sd.items()}
    torch.save(dict(score_model_state_dict=ren, epoch=1, steps=1), ck)
    with pytest.raises(RuntimeError, match="do not match the schema"):
        A.get_models(d, "train_configs.yaml", "task_configs.yaml", ck, "cpu", n_warmups=0)
    sd.pop("score_head.key_tensor_field.gnn_block_init.ga.alpha_dot")
    torch.save(dict(score_model_state_dict=sd, epoch=1, steps=1), ck)
    with pytest.raises(RuntimeError, match="Missing key"):
        A.get_models(d, "train_configs.yaml", "task_configs.yaml", ck, "cpu", n_warmups=0, strict_load=True)
    with pytest.raises(RuntimeError, match="do not match the schema"):
        A.get_models(d, "train_configs.yaml", "task_configs.yaml", ck, "cpu", n_warmups=0)
    sd2 = {k: v for k, v in sd_ref.items()}
    sd2["score_head.some_renamed_block.weight"] = torch.zeros(4)
    torch.save(dict(score_model_state_dict=sd2, epoch=1, steps=1), ck)
    with pytest.raises(RuntimeError, match="do not match the schema"):
        A.get_models(d, "train_configs.yaml", "task_configs.yaml", ck, "cpu", n_warmups=0)


def test_model_assembly_errors_match_reference():
    doc = _model_yaml(synthetic.score_head_kwargs(2))["model_kwargs"]
    bad = copy.deepcopy(doc); bad["key_kwargs"]["feature_extractor_name"] = "Nope"
    with pytest.raises(ValueError, match="Unknown feature extractor name: Nope"):
        A.MultiscaleScoreModel(**bad)
    bad = copy.deepcopy(doc); bad["query_model"] = "Nope"
    with pytest.raises(ValueError, match="Unknown query model: Nope"):
        A.MultiscaleScoreModel(**bad)
    bad = copy.deepcopy(doc); bad["score_head_kwargs"]["key_tensor_field_kwargs"]["irreps_input"] = "64x0e"
    with pytest.raises(AssertionError):               # multiscale_score_model.py:80
        A.MultiscaleScoreModel(**bad)
    bad = copy.deepcopy(doc); del bad["score_head_kwargs"]["lin_mult"]
    with pytest.raises(NotImplementedError):          # :68-70
        A.MultiscaleScoreModel(**bad)
    src = copy.deepcopy(doc)
    A.MultiscaleScoreModel(**src)
    assert src == doc                                 # the caller's dict is left alone


def test_agent_sample_argument_contract():
    class Dummy:
        diffusion_schedules = [[1.0, 0.1]]
    ag = A.DiffusionEdfAgent(models=[Dummy(), Dummy()])
    T = torch.zeros(3, 7)
    with pytest.raises(AssertionError, match="2 != 1"):
        ag.sample(None, None, T, [[1]], [[0.1], [0.1]], [1.0, 1.0])
    with pytest.raises(AssertionError, match="2 != 3"):
        ag.sample(None, None, T, [[1], [1]], [[0.1], [0.1]], [1.0, 1.0, 1.0])
    with pytest.raises(AssertionError, match=r"torch.Size\(\[3, 6\]\)"):
        ag.sample(None, None, torch.zeros(3, 6), [[1], [1]], [[0.1], [0.1]], [1.0, 1.0])
    with pytest.raises(AttributeError, match="unknown preprocess step"):          # the reference's getattr(preprocess, name) fails likewise
        A.DiffusionEdfAgent(models=[], preprocess_config=[dict(name="no_such_proc", kwargs={})])


# ---- task front-end: agent.yaml / server.yaml / preprocess.yaml ---------------------------------------------------------------

def test_preprocess_procs_of_the_shipped_yaml():
    """downsample (voxel average, metres) -> rescale (x100) as configs/*/preprocess.yaml lists them; poses only feel the rescale"""
    from diffusion_edf_amd import preprocess as PP
    x = torch.tensor([[0.001, 0.002, 0.003], [0.004, 0.006, 0.009], [0.011, 0.002, 0.003], [-0.004, 0.0, 0.0], [0.002, 0.001, 0.0]])
    f = torch.arange(15, dtype=torch.float32).reshape(5, 3)
    pcd = FeaturedPoints(x=x, f=f, b=torch.zeros(5, dtype=torch.long))
    cfg = [dict(name="downsample", kwargs=dict(voxel_size=0.01, coord_reduction="average")), dict(name="rescale", kwargs=dict(rescale_factor=100.0))]
    fn = PP.compose_proc_fn(cfg)
    out = fn(pcd)
    # voxels (ix,iy,iz): points 0,1,4 -> (0,0,0); point 2 -> (1,0,0); point 3 -> (-1,0,0); output sorted by voxel index
    assert out.x.shape == (3, 3) and out.f.shape == (3, 3) and torch.equal(out.b, torch.zeros(3, dtype=torch.long))
    assert torch.allclose(out.x[0], x[3] * 100) and torch.allclose(out.x[2], x[2] * 100)
    assert torch.allclose(out.x[1], x[[0, 1, 4]].mean(0) * 100) and torch.allclose(out.f[1], f[[0, 1, 4]].mean(0))
    T = torch.tensor([[1., 0, 0, 0, 0.1, 0.2, 0.3]])
    assert torch.allclose(fn(T), torch.tensor([[1., 0, 0, 0, 10., 20., 30.]])) and torch.equal(T[0, 4:], torch.tensor([0.1, 0.2, 0.3]))
    assert fn("anything else") == "anything else"
    back = PP.compose_proc_fn([dict(name="rescale", kwargs=dict(rescale_factor=0.01))])
    assert torch.allclose(back(fn(T)), T)
    centre = PP.downsample(pcd, voxel_size=0.01, coord_reduction="center")
    assert torch.allclose(centre.x, torch.tensor([[-0.005, 0.005, 0.005], [0.005, 0.005, 0.005], [0.015, 0.005, 0.005]]))
    crop = PP.crop_bbox(pcd, bbox=[[0.0, 0.02], [0.0, 0.01], [0.0, 0.01]])
    assert len(crop.x) == 4
    # a crop that names its targets (the sapien task files: ['scene_pcd']) refuses to guess what an unlabelled cloud is
    box = dict(bbox=[[0.0, 0.02], [0.0, 0.01], [0.0, 0.01]], targets=["scene_pcd"])
    assert len(PP.crop_bbox(pcd, role="scene_pcd", **box).x) == 4 and len(PP.crop_bbox(pcd, role="grasp_pcd", **box).x) == len(pcd.x)
    with pytest.raises(ValueError):
        PP.crop_bbox(pcd, **box)
    # two clouds in one batch vector are downsampled separately
    two = FeaturedPoints(x=torch.cat([x, x]), f=torch.cat([f, f]), b=torch.tensor([0] * 5 + [1] * 5))
    assert torch.equal(PP.downsample(two, 0.01).b, torch.tensor([0, 0, 0, 1, 1, 1]))


REF_CONFIGS = "/root/reference/configs"


@pytest.mark.skipif(not os.path.isdir(REF_CONFIGS), reason="reference tree not present")
def test_task_front_end_reads_every_shipped_task_directory():
    """agent.yaml / server.yaml / preprocess.yaml of all five shipped task directories (reference agent_server.py:48-86, 204-213)"""
    from diffusion_edf_amd import configs as CF
    dirs = CF.list_task_dirs(REF_CONFIGS)
    assert [os.path.basename(d) for d in dirs] == ["panda_bottle", "panda_bowl", "panda_mug", "sapien", "sapien_bottle"]
    for d in dirs:
        tc = CF.TaskConfigs.load(d)
        assert tc.root == "/root/reference" and tc.device == "cuda"
        for task in CF.TASKS:
            assert len(tc.models[task]) == 2                                            # low-res -> high-res cascade
            for e in tc.models[task] + ([tc.critic[task]] if tc.critic[task] else []):
                r = e.resolved(tc.root, "skip")
                assert os.path.isfile(os.path.join(r["configs_root_dir"], r["train_configs_file"])) and r["checkpoint_dir"] is None
                assert e.resolved(tc.root, "if_present")["checkpoint_dir"] is None        # every *.pt here is a git-LFS pointer
            kw = tc.sample_kwargs(task)
            assert len(kw["N_steps_list"]) == len(kw["timesteps_list"]) == len(kw["temperatures_list"]) == len(kw["diffusion_schedules_list"]) == 2
            assert tc.n_denoising_steps(task) in (650, 900, 1000) or tc.n_denoising_steps(task) > 0
        names = [p["name"] for p in tc.preprocess_config]
        assert names == (["crop_bbox"] if "sapien" in d else []) + ["downsample", "rescale"] and tc.preprocess_config[-1]["kwargs"]["rescale_factor"] == 100.0
        if "sapien" in d:                      # the crop targets the scene cloud only
            from diffusion_edf_amd import preprocess as PP
            fn = PP.compose_proc_fn(tc.preprocess_config)
            far = FeaturedPoints(x=torch.tensor([[5.0, 5.0, 5.0], [0.0, 0.0, 0.9]]), f=torch.zeros(2, 3), b=torch.zeros(2, dtype=torch.long))
            assert len(fn(far, role="scene_pcd").x) == 1 and len(fn(far, role="grasp_pcd").x) == 2
        assert [p["name"] for p in tc.unprocess_config] == ["rescale"]
    mug = CF.TaskConfigs.load(os.path.join(REF_CONFIGS, "panda_mug"))
    assert mug.critic["pick"] is not None and mug.n_denoising_steps("pick") == 900       # 200+200 | 200+200+100 (server.yaml:2)
    with pytest.raises(FileNotFoundError, match="git-LFS pointer"):
        mug.models["pick"][0].resolved(mug.root, "require")
    with pytest.raises(ValueError, match="Unknown task name"):
        mug.sample_kwargs("push")


@pytest.mark.skipif(not os.path.isdir(REF_CONFIGS), reason="reference tree not present")
def test_task_front_end_builds_the_agent_of_the_notebook_call_sequence():
    """configs/panda_mug -> DiffusionEdfAgent (two score models + critic built from the reference's own YAML, seeded init because the
    checkpoints are LFS pointers), pre-processing composed from preprocess.yaml, and the server.yaml arguments accepted by `sample`
    (the models are swapped for CPU stand-ins here: the heads themselves only run on the GPU)"""
    from diffusion_edf_amd import configs as CF
    tc = CF.TaskConfigs.load(os.path.join(REF_CONFIGS, "panda_mug"))
    feats = A.PrecomputedFeatures(None, "64x0e+32x1e+16x2e")
    ag = tc.build_agent("pick", extractors=[dict(key_extractor=feats), dict(key_extractor=feats)], critic_extractors=dict(key_extractor=feats),
                        device="cpu", checkpoints="skip", n_warmups=0)
    assert len(ag.models) == 2 and ag.critic is not None
    assert [type(m).__name__ for m in ag.models] == ["MultiscaleScoreModel"] * 2 and ag.critic.score_head.cfg.ebm
    assert ag.models[0].score_head.cfg.radii == [5.0, 10.0, 20.0, None] and ag.models[1].score_head.cfg.radii == [3.5, 5.0, 6.5, 8.0]
    assert ag.models[0].diffusion_schedules == [[1.0, 0.15], [0.15, 0.01]] or len(ag.models[0].diffusion_schedules) >= 1
    pcd = FeaturedPoints(x=torch.rand(200, 3) * 0.05, f=torch.rand(200, 3), b=torch.zeros(200, dtype=torch.long))
    out = ag.proc_fn(pcd)
    assert len(out.x) < 200 and float(out.x.max()) > 1.0                                  # voxel-downsampled, now in centimetres
    assert torch.allclose(ag.unprocess_fn(ag.proc_fn(torch.tensor([[1., 0, 0, 0, 0.1, 0.2, 0.3]]))), torch.tensor([[1., 0, 0, 0, 0.1, 0.2, 0.3]]))

    class Stage:                                   # stands in for ScoreModelBase.sample on CPU: one pose row per step
        def __init__(self, m):
            self.diffusion_schedules = m.diffusion_schedules
        get_key_pcd_multiscale = staticmethod(lambda p: None)
        get_query_pcd = staticmethod(lambda p: None)

        def sample(self, T_seed=None, N_steps=None, **kw):
            return T_seed[None].repeat(sum(N_steps) + 2, 1, 1)
    ag.models = [Stage(m) for m in ag.models]
    ag.critic = None
    T0 = torch.tensor([[1., 0, 0, 0, 0.1, 0.2, 0.3]] * 3)
    Ts, scene, grasp = ag.sample(pcd, pcd, T0, **tc.sample_kwargs("pick"))
    assert Ts.shape == (400 + 2 + 500 + 2, 3, 7) and torch.allclose(Ts[-1, :, 4:], T0[:, 4:] * 100.0)


def test_point_attentive_model_wraps_its_single_key_cloud():
    """point_attentive_score_model.py:106-107 returns [key_model(pcd)]; FeaturedPoints is a NamedTuple, so "is it already a sequence" must not be
    asked with isinstance(·, tuple)"""
    hk = synthetic.score_head_kwargs(2, radii=(None,))
    doc = _model_yaml(hk)["model_kwargs"]
    cloud = FeaturedPoints(x=torch.zeros(5, 3), f=torch.zeros(5, 240), b=torch.zeros(5, dtype=torch.long), w=torch.ones(5))
    m = A.PointAttentiveScoreModel(**doc, key_extractor=A.PrecomputedFeatures(cloud, "64x0e+32x1e+16x2e"))
    out = m.get_key_pcd_multiscale(None)
    assert isinstance(out, list) and len(out) == 1 and out[0] is cloud
    m2 = A.PointAttentiveScoreModel(**doc, key_extractor=A.PrecomputedFeatures([cloud], "64x0e+32x1e+16x2e"))
    assert len(m2.get_key_pcd_multiscale(None)) == 1


def _write_task_tree(root, schedules=((1.0, 0.15),)):
    """a task directory in the reference's layout (configs/<task>/{agent,server,preprocess}.yaml + one directory per model) for a place task:
    low-res and high-res score models with UNet key model + KeypointExtractor query model, and an EBM critic"""
    from test_keypoint_extractor import _query_kwargs
    task = os.path.join(root, "configs", "toy_place")

    def model_dir(name, radii, ebm):
        hk = synthetic.score_head_kwargs(2, radii=radii)
        if ebm:
            hk.update(ebm=True, edge_time_encoding=False, query_time_encoding=False)
        doc = _model_yaml(hk)
        doc["model_kwargs"]["query_model"] = "KeypointExtractor"
        doc["model_kwargs"]["query_kwargs"] = _query_kwargs((5.0, 10.0, 20.0, 40.0), bbox=None)
        d = os.path.join(task, name)
        os.makedirs(d, exist_ok=True)
        yaml.safe_dump(dict(model_config_file='score_model_configs.yaml', device='cuda:0',
                            diffusion_configs=dict(time_schedules=[list(s) for s in schedules], t_augment=0.01)), open(os.path.join(d, 'train_configs.yaml'), 'w'))
        yaml.safe_dump(dict(task_type='place'), open(os.path.join(d, 'task_configs.yaml'), 'w'))
        yaml.safe_dump(doc, open(os.path.join(d, 'score_model_configs.yaml'), 'w'))
        return dict(configs_root_dir=os.path.join("configs", "toy_place", name), train_configs_file="train_configs.yaml",
                    task_configs_file="task_configs.yaml", checkpoint_dir=None, n_warmups=0)
    agent = dict(device="cuda:0", model_kwargs=dict(
        place_models_kwargs=[model_dir("place_lowres", (5., 10., 20., None), False), model_dir("place_highres", (3.5, 5., 6.5, 8.), False)],
        place_critic_kwargs=model_dir("place_ebm", (5., 10., 20., None), True), pick_models_kwargs=[], pick_critic_kwargs=None))
    os.makedirs(task, exist_ok=True)
    yaml.safe_dump(agent, open(os.path.join(task, "agent.yaml"), "w"))
    yaml.safe_dump(dict(place_diffusion_configs=dict(N_steps_list=[[3], [2]], timesteps_list=[[0.04], [0.02]], temperatures_list=[1.0, 1.0],
                                                     diffusion_schedules_list=[[[1.0, 0.3]], [[0.3, 0.1]]], log_t_schedule=True,
                                                     time_exponent_temp=1.0, time_exponent_alpha=0.5)), open(os.path.join(task, "server.yaml"), "w"))
    yaml.safe_dump(dict(preprocess_config=[dict(name="downsample", kwargs=dict(voxel_size=0.005, coord_reduction="average")),
                                           dict(name="rescale", kwargs=dict(rescale_factor=100.0))],
                        unprocess_config=[dict(name="rescale", kwargs=dict(rescale_factor=0.01))]), open(os.path.join(task, "preprocess.yaml"), "w"))
    return task


def test_toy_task_tree_loads_on_the_host(tmp_path):
    from diffusion_edf_amd import configs as CF
    tc = CF.TaskConfigs.load(_write_task_tree(str(tmp_path)))
    assert len(tc.models["place"]) == 2 and tc.critic["place"] is not None and tc.n_denoising_steps("place") == 5


@pytest.mark.gpu
def test_task_directory_to_ranked_poses_on_the_gpu(tmp_path):
    """the whole deployment path with nothing injected: task directory -> DiffusionEdfAgent (UNet + KeypointExtractor + score heads + critic, all
    on the HIP path) -> sample(raw clouds in metres, T_seed) -> trajectories ordered by critic energy.  Every stage has its own parity test;
    this one checks that they are wired like agent.py:98-186: shapes, units, ordering, determinism."""
    from diffusion_edf_amd import configs as CF
    dev = torch.device("cuda:0")
    tc = CF.TaskConfigs.load(_write_task_tree(str(tmp_path)))
    ag = tc.build_agent("place", checkpoints="skip", device="cuda:0", n_warmups=0)
    from diffusion_edf_amd.keypoint_extractor import KeypointExtractor
    from diffusion_edf_amd.unet import UnetFeatureExtractor
    assert all(type(m.key_model) is UnetFeatureExtractor and type(m.query_model) is KeypointExtractor for m in ag.models + [ag.critic])
    import numpy as np
    g = torch.Generator().manual_seed(0)
    scene_x = torch.from_numpy(synthetic.make_scene(6000, seed=1).astype(np.float32)) * 0.01                   # metres
    grasp_x = torch.from_numpy(synthetic.make_grasp(1500, seed=2).astype(np.float32)) * 0.01
    fp = lambda x: FeaturedPoints(x=x.to(dev), f=torch.rand(len(x), 3, generator=g).to(dev), b=torch.zeros(len(x), dtype=torch.long, device=dev))
    scene, grasp = fp(scene_x), fp(grasp_x)
    T0 = synthetic.make_poses(12, seed=3).to(dev)
    T0[:, 4:] *= 0.01                                                                                          # metres
    kw = tc.sample_kwargs("place")
    torch.manual_seed(7)             # get_models builds the extractors with deterministic=False like agent.py:20-64: FPS starts at a random point
    Ts, scene_p, grasp_p, info = ag.sample(scene, grasp, T0, **kw, return_info=True, seed=5)
    assert Ts.shape == ((3 + 2) + (2 + 2), 12, 7) and bool(torch.isfinite(Ts).all())
    assert len(scene_p.x) < len(scene.x) and float(scene_p.x.abs().max()) > 5.0                                # downsampled, centimetres
    assert torch.allclose(Ts[..., :4].norm(dim=-1), torch.ones_like(Ts[..., 0]), atol=1e-5)
    e = info["energy"]
    assert e.shape == (12,) and bool((e[1:] >= e[:-1]).all()) and float(e[-1] - e[0]) > 0
    first = Ts[0].double()
    assert float((first[:, 4:].sort(dim=0).values - (T0[:, 4:].double() * 100.0).sort(dim=0).values).abs().max()) < 1e-4   # the seeds, in cm, re-ordered
    assert float((Ts[-1, :, 4:] - Ts[0, :, 4:]).abs().max()) > 1e-3                                           # the poses moved
    torch.manual_seed(7)
    Ts2, _, _ = ag.sample(scene, grasp, T0, **kw, seed=5)
    assert torch.equal(Ts2, Ts)                                         # same FPS starts + counter-based Langevin noise: bit-reproducible
    back = ag.unprocess_fn(Ts[-1])
    assert torch.allclose(back[:, 4:], Ts[-1][:, 4:] * 0.01)
