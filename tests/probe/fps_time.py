"""time dedf_fps alone: python tests/probe/fps_time.py   (DEDF_FPS_BUCKETED=0: the exhaustive kernel everywhere; default: bucketed + batched from 1 025 to 16 384 points)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from diffusion_edf_amd import synthetic
from diffusion_edf_amd.connectivity import fps
dev = torch.device("cuda:0")
sizes = ((16384, 0.2), (16384, 0.05), (8192, 0.2), (4096, 0.25), (3277, 0.2), (2560, 0.2), (2048, 0.2), (1536, 0.2), (1100, 0.2), (656, 0.2), (65536, 0.05))
print("DEDF_FPS_BUCKETED =", os.environ.get("DEDF_FPS_BUCKETED", "(default)"))
for n, ratio in sizes:
    x = torch.from_numpy(synthetic.make_scene(n, seed=0).astype(np.float32)).to(dev)
    b = torch.zeros(n, dtype=torch.long, device=dev)
    fps(x, b, ratio=ratio, random_start=False); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        idx = fps(x, b, ratio=ratio, random_start=False)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    print(f"fps n={n} -> {len(idx)} samples: {ms:.3f} ms = {ms * 1e3 / len(idx):.2f} us / sample")
