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
                                  key_kwargs=dict(feature_extractor_name='UnetFeatureExtractor', feature_extractor_kwargs=dict(irreps_output=irr)),
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
    with pytest.raises(NotImplementedError):          # extractors are injected, never silently replaced
        m.get_key_pcd_multiscale(None)
    if doc["model_kwargs"]["query_model"] == "StaticKeypointModel":       # parameters only: part of the model (keypoint_extractor.py:22-47)
        qk = doc["model_kwargs"]["query_kwargs"]
        from diffusion_edf_amd.gnn_data import FeaturedPoints
        q = m.get_query_pcd(FeaturedPoints(x=torch.zeros(5, 3), f=torch.zeros(5, 3), b=torch.zeros(5, dtype=torch.long)))
        nK = len(qk["keypoint_coords"])
        assert q.x.shape == (nK, 3) and q.f.shape == (nK, m.score_head.cfg.dim) and q.w.shape == (nK,) and q.b.shape == (nK,)
        assert torch.equal(q.x, torch.tensor(qk["keypoint_coords"])) and bool(((q.w > 0) & (q.w < 1)).all())
        assert {k for k in m.state_dict() if k.startswith("query_model.")} == {
            "query_model.keypoint_coords", "query_model.keypoint_features", "query_model.keypoint_weights"}
    else:
        with pytest.raises(NotImplementedError):
            m.get_query_pcd(None)


def test_checkpoint_load_like_the_reference_agent(tmp_path):
    kw = synthetic.score_head_kwargs(2)
    d = str(tmp_path / "pick_lowres")
    _write_config_dir(d, kw, schedules=((1.0, 0.15), (0.15, 0.01)))
    cfg = params.HeadConfig.from_kwargs(kw)
    P = params.init_params(cfg, seed=11, randomize_all=True)
    sd = {"score_head." + k: v for k, v in P.items()}
    sd["key_model.blocks.0.weight"] = torch.zeros(3)             # extractor weights of the full reference model
    sd["query_model.keypoint_coords"] = torch.tensor([[0.5, 0.5, 10.5], [-0.5, -0.5, 10.5]])
    sd["query_model.keypoint_features"] = torch.randn(2, cfg.dim)
    sd["query_model.keypoint_weights"] = torch.tensor([0.3, -1.2])
    ck = str(tmp_path / "Pick_LowRes_300.pt")
    torch.save(dict(score_model_state_dict=sd, epoch=300, steps=12345), ck)
    m = A.get_models(d, "train_configs.yaml", "task_configs.yaml", ck, "cpu", n_warmups=0)          # strict_load=False: agent.py:28
    got = m.score_head.state_dict()
    assert set(got) == set(P)
    for k, v in P.items():
        assert torch.equal(got[k], v), k
    q = m.get_query_pcd(synthetic.make_query(cfg, 0, seed=0, static_keypoints=True))        # StaticKeypointModel: weights from the checkpoint
    assert torch.equal(q.f, sd["query_model.keypoint_features"]) and torch.allclose(q.w, torch.sigmoid(torch.tensor([0.3, -1.2])))
    assert m.diffusion_schedules == [[1.0, 0.15], [0.15, 0.01]]
    with pytest.raises(RuntimeError, match="Unexpected key"):
        A.get_models(d, "train_configs.yaml", "task_configs.yaml", ck, "cpu", n_warmups=0, strict_load=True)
    sd.pop("score_head.key_tensor_field.gnn_block_init.ga.alpha_dot")
    torch.save(dict(score_model_state_dict={k: v for k, v in sd.items() if k.startswith("score_head.")}, epoch=1, steps=1), ck)
    with pytest.raises(RuntimeError, match="Missing key"):
        A.get_models(d, "train_configs.yaml", "task_configs.yaml", ck, "cpu", n_warmups=0, strict_load=True)


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
    with pytest.raises(NotImplementedError):
        A.DiffusionEdfAgent(models=[], preprocess_config=[dict(name="downsample")])
