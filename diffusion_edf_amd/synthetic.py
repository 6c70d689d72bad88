"""Seeded synthetic inputs of the shapes BASELINE.md names.

No demo clouds or checkpoints exist in the reference tree (LFS stubs), so bench.py and the tests use these
generators.  Units are centimetres (reference README.md:82).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from .gnn_data import FeaturedPoints
from .params import HeadConfig
from .so3 import irreps_dim


def ebm_head_kwargs(lmax: int = 2, radii=(3.5, 5., 6.5, 8.)) -> dict:
    """`score_head_kwargs` of the critic: reference configs/panda_mug/pick_ebm/score_model_configs.yaml:3-26 (no time encoding)"""
    kw = score_head_kwargs(lmax, radii)
    kw.update(ebm=True, edge_time_encoding=False, query_time_encoding=False)
    return kw


def score_head_kwargs(lmax: int = 2, radii=(5., 10., 20., None), query_time_encoding: bool = False, edge_time_encoding: bool = True) -> dict:
    """The `score_head_kwargs` block of reference configs/panda_mug/pick_lowres/score_model_configs.yaml:3-25
    (+ the keys multiscale_score_model.py:79-85 injects), with irreps truncated at `lmax`."""
    irr = '+'.join(['64x0e', '32x1e', '16x2e', '8x3e'][:lmax + 1])
    sh = '+'.join(['1x0e', '1x1e', '1x2e', '1x3e'][:lmax + 1])
    return dict(
        max_time=1., time_emb_mlp=[256, 128, 64], ang_mult=2.5, lin_mult=15.,
        edge_time_encoding=edge_time_encoding, query_time_encoding=query_time_encoding,
        key_tensor_field_kwargs=dict(
            irreps_input=irr, irreps_output=irr, irreps_sh=sh, num_heads=4, fc_neurons=[-1, 128, 64],
            length_emb_dim=64, r_cluster_multiscale=list(radii), n_layers=1, irreps_mlp_mid=3,
            cutoff_method='edge_attn', r_mincut_nonscalar_sh=0.3,
            length_enc_max_r=100. if radii[-1] is None else None, use_src_point_attn=False),
        irreps_query_edf=irr)


def fps(x: np.ndarray, ratio: float) -> np.ndarray:
    """Farthest point sampling, start index 0, ceil(ratio*N) points (torch_cluster.fps with
    random_start=False, as reference connectivity.py:62)."""
    N = x.shape[0]
    k = int(math.ceil(ratio * N))
    idx = np.zeros(k, dtype=np.int64)
    d = np.full(N, np.inf)
    cur = 0
    for i in range(k):
        idx[i] = cur
        d = np.minimum(d, ((x - x[cur]) ** 2).sum(-1))
        cur = int(np.argmax(d))
    return idx


def make_scene(n_points: int, seed: int = 0) -> np.ndarray:
    """60 % plane patch [-25,25]^2 x {0}, 40 % vertical cylinder surface (r=4, h=10, centre (0,0,5))."""
    rng = np.random.default_rng(seed)
    n_plane = int(round(0.6 * n_points))
    n_cyl = n_points - n_plane
    plane = np.stack([rng.uniform(-25, 25, n_plane), rng.uniform(-25, 25, n_plane), np.zeros(n_plane)], -1)
    th = rng.uniform(0, 2 * np.pi, n_cyl)
    cyl = np.stack([4 * np.cos(th), 4 * np.sin(th), rng.uniform(0, 10, n_cyl)], -1)
    return np.concatenate([plane, cyl], 0)


def make_grasp(n_points: int, seed: int = 0) -> np.ndarray:
    """uniform on the surface of a 6x6x10 cm box centred (0,0,8) in the gripper frame."""
    rng = np.random.default_rng(seed + 1000)
    ext = np.array([6., 6., 10.])
    areas = np.array([ext[1] * ext[2], ext[0] * ext[2], ext[0] * ext[1]])
    face = rng.choice(3, size=n_points, p=areas / areas.sum())
    p = rng.uniform(-0.5, 0.5, size=(n_points, 3)) * ext
    sign = rng.choice([-0.5, 0.5], size=n_points)
    p[np.arange(n_points), face] = sign * ext[face]
    return p + np.array([0., 0., 8.])


def make_key_clouds(cfg: HeadConfig, n_scene: int, seed: int = 0, ratio: float = 0.2,
                    dtype=torch.float32) -> List[FeaturedPoints]:
    """FPS cascade (ratio 0.2, n_scales times) of the synthetic scene; features ~ N(0,1) per component."""
    x = make_scene(n_scene, seed)
    g = torch.Generator().manual_seed(seed + 17)
    out = []
    for n in range(cfg.n_scales):
        x = x[fps(x, ratio)]
        f = torch.randn(x.shape[0], cfg.dim, generator=g, dtype=torch.float64)
        out.append(FeaturedPoints(x=torch.tensor(x, dtype=dtype), f=f.to(dtype),
                                  b=torch.zeros(x.shape[0], dtype=torch.long)))
    return out


def make_query(cfg: HeadConfig, n_grasp: int, seed: int = 0, ratio: float = 0.1, static_keypoints: bool = False,
               dtype=torch.float32) -> FeaturedPoints:
    g = torch.Generator().manual_seed(seed + 29)
    if static_keypoints:      # reference configs/panda_mug/pick_lowres/score_model_configs.yaml:76-80
        x = np.array([[0.5, 0.5, 10.5], [-0.5, -0.5, 10.5]])
    else:
        x = make_grasp(n_grasp, seed)
        x = x[fps(x, ratio)]
    f = torch.randn(x.shape[0], cfg.dim, generator=g, dtype=torch.float64)
    w = torch.sigmoid(torch.randn(x.shape[0], generator=g, dtype=torch.float64))
    return FeaturedPoints(x=torch.tensor(x, dtype=dtype), f=f.to(dtype), b=torch.zeros(x.shape[0], dtype=torch.long),
                          w=w.to(dtype))


def make_poses(n: int, seed: int = 1, first_pose_index: int = 0, near_object: bool = False,
               dtype=torch.float64) -> torch.Tensor:
    """(n,7) [qw,qx,qy,qz,x,y,z]; q = normalised N(0,1)^4 standardised to w>=0 (reference
    transforms.py:349-355), x ~ U([-20,20]^2 x [0,25]).  Each pose is drawn from its own generator keyed by
    (seed, global pose index) so that a pose-sharded run sees exactly the poses of the single-GPU run."""
    out = np.zeros((n, 7))
    for i in range(n):
        rng = np.random.default_rng([seed, first_pose_index + i])
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        if q[0] < 0:
            q = -q
        if near_object:
            p = np.array([rng.uniform(-8, 8), rng.uniform(-8, 8), rng.uniform(0, 15)])
        else:
            p = np.array([rng.uniform(-20, 20), rng.uniform(-20, 20), rng.uniform(0, 25)])
        out[i, :4], out[i, 4:] = q, p
    return torch.tensor(out, dtype=dtype)


# ---- the feature-extractor blocks of the shipped YAML files (tests, bench) --------------------------------------------------------

def unet_kwargs(kind: str = "panda_lowres") -> dict:
    """the four feature_extractor_kwargs blocks the reference ships (configs/*/*/score_model_configs.yaml)"""
    narrow, wide = "32x0e+16x1e+8x2e", "64x0e+32x1e+16x2e"
    sh = "1x0e+1x1e+1x2e"
    if kind.endswith("_lmax3"):        # BASELINE config 5: the same networks with one more degree (64x0e+32x1e+16x2e+8x3e, SH up to 3e); no shipped YAML uses it
        kw = unet_kwargs(kind[:-len("_lmax3")])
        up = lambda s_: {narrow: narrow + "+4x3e", wide: wide + "+8x3e", sh: sh + "+1x3e"}[s_]
        return dict(kw, irreps_output=up(kw["irreps_output"]), irreps_emb=[up(i) for i in kw["irreps_emb"]], irreps_edge_attr=[up(i) for i in kw["irreps_edge_attr"]])
    base = dict(irreps_input="3x0e", irreps_output=wide, irreps_mlp_mid=3, attn_type="mlp", alpha_drop=0.1, proj_drop=0.1, drop_path_rate=0.0,
                pool_method="fps", n_layers_midstream=2)
    if kind == "panda_lowres":         # pool_ratio 0.2 (…/pick_lowres, place_ebm …)
        return dict(base, irreps_emb=[narrow, narrow, wide, wide], fc_neurons=[[32, 16, 16]] * 2 + [[64, 32, 32]] * 2, irreps_edge_attr=[sh] * 4,
                    num_heads=[4] * 4, n_layers=[2] * 4, pool_ratio=[0.2] * 4, radius=[3.0, None, None, None], n_scales=4)
    if kind == "panda_highres":        # pool_ratio 0.25
        return dict(unet_kwargs("panda_lowres"), pool_ratio=[0.25] * 4)
    if kind == "sapien_lowres":        # every level wide
        return dict(unet_kwargs("panda_highres"), irreps_emb=[wide] * 4, fc_neurons=[[64, 32, 32]] * 4)
    if kind == "sapien_highres":       # one scale, seven layers
        return dict(base, irreps_emb=[wide], fc_neurons=[[64, 32, 32]], irreps_edge_attr=[sh], num_heads=[4], n_layers=[7], pool_ratio=[0.25],
                    radius=[3.0], n_scales=1)
    raise KeyError(kind)


def keypoint_extractor_kwargs(radii=(5.0, 10.0, 20.0, 40.0), bbox=((-30.0, 30.0), (-30.0, 30.0), (8.0, 100.0)), unet="panda_highres", pool_ratio=0.1):
    """the query_kwargs block of configs/panda_*/place_*/score_model_configs.yaml (``unet`` ending in ``_lmax3``: the same with one more degree)"""
    if unet.endswith("_lmax3"):
        kw = keypoint_extractor_kwargs(radii, bbox, unet[:-len("_lmax3")], pool_ratio)
        kw["feature_extractor_kwargs"] = unet_kwargs(unet)
        kw["tensor_field_kwargs"] = dict(kw["tensor_field_kwargs"], irreps_output='64x0e+32x1e+16x2e+8x3e', irreps_sh="1x0e+1x1e+1x2e+1x3e")
        return kw
    return dict(weight_activation="sigmoid", weight_mult=None,
                keypoint_kwargs=dict(pool_ratio=pool_ratio, weight_pre_emb_dim=64, bbox=None if bbox is None else [list(b) for b in bbox]),
                feature_extractor_kwargs=unet_kwargs(unet),
                tensor_field_kwargs=dict(irreps_output='64x0e+32x1e+16x2e', irreps_sh="1x0e+1x1e+1x2e", num_heads=4, fc_neurons=[-1, 32, 32], length_emb_dim=64,
                                         r_cluster_multiscale=list(radii), n_scales=len(radii)))


def config5_model_kwargs(radii=(5., 10., 20., None), field_radii=(5.0, 10.0, 20.0, 40.0)) -> dict:
    """BASELINE config 5 as ONE model: the ``model_kwargs`` block of a ``score_model_configs.yaml`` in the reference's schema
    (configs/panda_mug/place_lowres/score_model_configs.yaml: UNet key model, KeypointExtractor query model, MultiscaleScoreModel head) with one
    more degree everywhere -- irreps 64x0e+32x1e+16x2e+8x3e (narrow UNet levels 32x0e+16x1e+8x2e+4x3e), SH up to 3e.  No shipped YAML uses
    lmax 3; the shapes are the shipped ones plus the l = 3 block.  The keys multiscale_score_model.py:79-93 injects are absent, as in the file."""
    import copy
    sh = copy.deepcopy(score_head_kwargs(3, radii))
    sh.pop('irreps_query_edf')
    for k in ('irreps_input', 'use_src_point_attn'):
        sh['key_tensor_field_kwargs'].pop(k, None)
    return dict(score_head_kwargs=sh,
                key_kwargs=dict(feature_extractor_name='UnetFeatureExtractor', feature_extractor_kwargs=unet_kwargs("panda_lowres_lmax3")),
                query_model='KeypointExtractor',
                query_kwargs=keypoint_extractor_kwargs(field_radii, bbox=None, unet="panda_highres_lmax3"))
