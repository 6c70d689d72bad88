"""Randomised parity sweep of the feature extractors against the fp64 restatement (GPU box: `python tests/stress_extractors.py [n] [seed]`).
Every case draws the extractor (UNet shape / forward-only / KeypointExtractor with or without bbox), the cloud (size 30 ... 5000, scene- or
object-like, optionally scaled so that levels run out of neighbours) and the weights at random."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from diffusion_edf_amd import synthetic
from diffusion_edf_amd.gnn_data import FeaturedPoints
from oracle import restatement as R
from oracle import unet_oracle as U
from test_keypoint_extractor import _field_cfg, _object_cloud
from test_unet import _oracle_cfg, _randomized

WIDE = [(64, 0), (32, 1), (16, 2)]


def block_err(got, ref):
    worst, off = 0.0, 0
    for mul, l in WIDE:
        d = mul * (2 * l + 1)
        worst = max(worst, float((got[:, off:off + d] - ref[:, off:off + d]).abs().max()) / max(float(ref[:, off:off + d].abs().max()), 1e-3 * float(ref.abs().max())))
        off += d
    return worst


def run_case(i, rng):
    from diffusion_edf_amd.keypoint_extractor import KeypointExtractor
    from diffusion_edf_amd.unet import ForwardOnlyFeatureExtractor, UnetFeatureExtractor
    dev = torch.device("cuda:0")
    what = ["unet", "unet", "forward_only", "keypoint"][int(rng.integers(0, 4))]
    kind = ["panda_lowres", "panda_highres", "sapien_lowres", "sapien_highres"][int(rng.integers(0, 4))]
    n = int(rng.choice([int(rng.integers(30, 300)), int(rng.integers(300, 2000)), int(rng.integers(2000, 5000))]))
    seed = int(rng.integers(0, 1 << 20))
    scale = float(rng.choice([1.0, 1.0, 0.3, 3.0]))                  # denser / sparser than the radii expect
    x = (synthetic.make_scene(n, seed=seed).astype(np.float32) if rng.integers(0, 2) else _object_cloud(n, seed=seed).numpy()) * np.float32(scale)
    x = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
    f = torch.rand(n, 3, generator=torch.Generator().manual_seed(seed))
    pcd = FeaturedPoints(x=x.to(dev), f=f.to(dev), b=torch.zeros(n, dtype=torch.long, device=dev), w=None)
    if what == "keypoint":
        radii = tuple(sorted(float(r) for r in rng.uniform(2.0, 30.0, size=4)))
        bbox = None if rng.integers(0, 2) else [[-1e3, 1e3], [-1e3, 1e3], [float(np.median(x[:, 2].numpy())), 1e3]]
        kw = synthetic.keypoint_extractor_kwargs(radii, bbox=bbox, unet=kind if kind != "sapien_highres" else "panda_lowres", pool_ratio=float(rng.choice([0.05, 0.1, 0.3])))
        m = KeypointExtractor(**kw, deterministic=True)
        sd = _randomized(m, seed=seed % 1000)
        xr, fr, wr = U.keypoint_extractor_forward(_oracle_cfg(m.feature_extractor), _field_cfg(radii), R.cast_params(sd, torch.float64), x, f.double(),
                                                  kw["keypoint_kwargs"]["pool_ratio"], bbox=bbox)
        out = m.to(dev)(pcd)
        ok = torch.equal(out.x.cpu(), xr)
        err = max(block_err(out.f.cpu().double(), fr), float((out.w.cpu().double() - wr).abs().max()))
        desc = f"keypoint[{kw['feature_extractor_kwargs']['pool_ratio'][0]}] radii {[round(r, 1) for r in radii]} bbox {bbox is not None} -> {len(xr)} key points"
    else:
        cls = ForwardOnlyFeatureExtractor if what == "forward_only" else UnetFeatureExtractor
        m = cls(**synthetic.unet_kwargs(kind), deterministic=True)
        sd = _randomized(m, seed=seed % 1000)
        ref = U.unet_forward(_oracle_cfg(m), R.cast_params(sd, torch.float64), x, f.double(), forward_only=what == "forward_only")
        out = m.to(dev)(pcd)
        ok = len(out) == len(ref) and all(torch.equal(o.x.cpu(), r[0]) for o, r in zip(out, ref))
        err = max(block_err(o.f.cpu().double(), r[1]) for o, r in zip(out, ref))
        desc = f"{what}[{kind}] levels {[len(r[0]) for r in ref]}"
    print(f"case {i:3d} err {err:.2e} coords_ok {ok}  n {n} scale {scale} {desc}", flush=True)
    return err, ok


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    res = [run_case(i, rng) for i in range(n)]
    bad = [i for i, (e, ok) in enumerate(res) if not (ok and e < 2e-4)]
    print("FAILED" if bad else "ALL OK", len(bad), "of", n, "largest error", max(e for e, _ in res), bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
