"""Every score-model config the reference ships (MultiscaleScoreModel and PointAttentiveScoreModel) must be accepted unchanged by the drop-in heads and by the C ABI.
Reads /root/reference/configs (build container only; skipped on boxes without the reference tree).  The score-head kwargs
are assembled exactly as reference multiscale_score_model.py:64-112 does (irreps_input / irreps_query_edf injected from
the key / query model outputs)."""
import ctypes as C
import glob
import os

import numpy as np
import pytest
import yaml

from diffusion_edf_amd import _lib, params

CFG_ROOT = "/root/reference/configs"
FILES = sorted(glob.glob(os.path.join(CFG_ROOT, "*", "*", "score_model_configs.yaml")))
pytestmark = pytest.mark.skipif(not FILES, reason="reference tree not present")


def _score_head_kwargs(doc):
    mk = doc["model_kwargs"]
    sh = dict(mk["score_head_kwargs"])
    tf = dict(sh["key_tensor_field_kwargs"])
    tf["irreps_input"] = mk["key_kwargs"]["feature_extractor_kwargs"]["irreps_output"]
    tf["use_src_point_attn"] = doc["model_name"] == "PointAttentiveScoreModel"          # point_attentive_score_model.py:71-72
    tf["use_dst_point_attn"] = False
    sh["key_tensor_field_kwargs"] = tf
    q = mk["query_kwargs"]
    sh["irreps_query_edf"] = q["irreps_output"] if "irreps_output" in q else q["feature_extractor_kwargs"]["irreps_output"]
    return sh


@pytest.mark.parametrize("path", FILES, ids=[os.path.relpath(f, CFG_ROOT) for f in FILES])
def test_shipped_config_is_accepted(path, built_lib):
    doc = yaml.safe_load(open(path))
    assert doc["model_name"] in ("MultiscaleScoreModel", "PointAttentiveScoreModel")
    kw = _score_head_kwargs(doc)
    cfg = params.HeadConfig.from_kwargs(kw)
    assert cfg.irreps == [(64, 0), (32, 1), (16, 2)] and cfg.num_heads == 4 and cfg.n_scales in (1, 4)
    assert cfg.ebm == bool(kw.get("ebm", False)) and cfg.use_src_point_attn == (doc["model_name"] == "PointAttentiveScoreModel")
    # [64 + time_emb, 128, 64]: 128 for time_emb_mlp [256,128,64], 192 for the sapien high-res [512,256,128]; 64 for the EBM critic;
    # the sapien place_* score heads have the narrow radial MLP [128, 32, 32]
    narrow = kw["key_tensor_field_kwargs"]["fc_neurons"][1:] == [32, 32]
    assert cfg.fc_neurons == ([64, 128, 64] if cfg.ebm else [64 + cfg.time_emb_mlp[-1]] + ([32, 32] if narrow else [128, 64]))
    cc = _lib.make_config(cfg, -1)
    names = _lib.param_names(cc)
    assert names == [(n, int(np.prod(s))) for n, s, _, _ in params.param_spec(cfg)]
    P = params.init_params(cfg, seed=1)
    blob = _lib.pack_params(cc, P)
    h = C.c_void_p()
    assert built_lib.dedf_create(C.byref(cc), blob.ctypes.data_as(C.POINTER(C.c_float)), blob.size, C.byref(h)) == _lib.OK
    built_lib.dedf_destroy(h)


def test_all_multiscale_configs_seen():
    kinds = [yaml.safe_load(open(f))["model_name"] for f in FILES]
    assert kinds.count("MultiscaleScoreModel") == 22 and kinds.count("PointAttentiveScoreModel") == 4
