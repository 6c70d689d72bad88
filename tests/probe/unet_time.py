"""Time the feature extractors on the GPU: UNet on a 16 384-point scene (BASELINE config 5's size, lmax 2), KeypointExtractor on a 4 096-point
grasp cloud.  Usage: python tests/probe/unet_time.py [n_scene] [reps]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from diffusion_edf_amd import synthetic                                        # noqa: E402
from diffusion_edf_amd.gnn_data import FeaturedPoints                          # noqa: E402
from diffusion_edf_amd.keypoint_extractor import KeypointExtractor             # noqa: E402
from diffusion_edf_amd.unet import UnetFeatureExtractor                        # noqa: E402
from test_keypoint_extractor import _object_cloud, _query_kwargs               # noqa: E402
from test_unet import _unet_kwargs                                             # noqa: E402

n_scene = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")


def timed(fn, reps):
    fn(); fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, out


fp = lambda x, f: FeaturedPoints(x=x.to(dev), f=f.to(dev), b=torch.zeros(len(x), dtype=torch.long, device=dev), w=None)
for kind in ("panda_lowres", "sapien_lowres", "panda_lowres_lmax3"):      # the last one: BASELINE config 5 as written (16 k points, lmax 3)
    m = UnetFeatureExtractor(**(synthetic.unet_kwargs(kind) if kind.endswith("_lmax3") else _unet_kwargs(kind)), deterministic=True).to(dev)
    x = torch.from_numpy(synthetic.make_scene(n_scene, seed=0).astype(np.float32))
    pcd = fp(x, torch.rand(n_scene, 3))
    t0 = time.perf_counter(); m(pcd); torch.cuda.synchronize()
    first = (time.perf_counter() - t0) * 1e3
    ms, out = timed(lambda: m(pcd), reps)
    print(f"UnetFeatureExtractor[{kind}] {n_scene} pts -> {[len(o.x) for o in out]}: {ms:.2f} ms / forward (first call incl. weight packing {first:.0f} ms)")
k = KeypointExtractor(**_query_kwargs(), deterministic=True).to(dev)
g = _object_cloud(4096, seed=3)
pcd = fp(g, torch.rand(len(g), 3))
ms, out = timed(lambda: k(pcd), reps)
print(f"KeypointExtractor 4096 pts -> {len(out.x)} key points: {ms:.2f} ms / forward")
