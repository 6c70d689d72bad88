"""timing of the graph primitives (dedf_fps / dedf_radius) at the sizes of BASELINE config 5 (16k-pt scene) and C2"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from diffusion_edf_amd import connectivity as K

def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n

for n in (4096, 16384, 65536):
    x = torch.from_numpy(np.random.default_rng(0).uniform(-25, 25, size=(n, 3)).astype(np.float32)).cuda()
    ms = t(lambda: K.fps(x, None, ratio=0.2, random_start=False))
    k = -(-n // 5)
    print(f"fps n={n} k={k}: {ms:.3f} ms = {ms * 1e3 / k:.2f} us per sample")
    idx = K.fps(x, None, ratio=0.2, random_start=False)
    y = x[idx]
    for r in (3.0, 6.0):
        e = K.radius(x, y, r, max_num_neighbors=1000)
        ms = t(lambda: K.radius(x, y, r, max_num_neighbors=1000))
        print(f"radius src={n} dst={len(y)} r={r}: {e.shape[1]} edges, {ms:.3f} ms ({n * len(y) / ms / 1e6:.1f} G pair tests/s incl. the host round trip)")
