"""Randomised bit-exactness sweep of dedf_fps (the bucketed + batched kernel, 1 025 ... 16 384 points) against the numpy oracle
(GPU box: `python tests/stress_fps.py [n_cases] [seed]`).  Every case draws the cloud size, the sampling ratio, the start point and the kind of
cloud: surfaces, Gaussian clusters of very different sizes, a jittered lattice (near ties), an exact lattice (hundreds of exact ties), copies
of a few points (minima reach 0 early), extreme aspect ratios, clouds far from the origin or scaled by 1e-4 / 1e4 (few significant bits in
the distances: more ties)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusion_edf_amd import _lib, synthetic
from oracle import graph_oracle as G


def draw_cloud(rng, n):
    kind = int(rng.integers(0, 9))
    if kind == 0:
        x = synthetic.make_scene(n, seed=int(rng.integers(0, 1 << 20)))
    elif kind == 1:
        x = rng.uniform(-1, 1, (n, 3)) * rng.uniform(0.1, 50, 3)
    elif kind == 2:                                                              # clusters of very different sizes and densities
        k = int(rng.integers(2, 9))
        c = rng.normal(0, 30, (k, 3)); s = 10.0 ** rng.uniform(-2, 1, k)
        a = rng.integers(0, k, n)
        x = c[a] + rng.normal(0, 1, (n, 3)) * s[a, None]
    elif kind == 3:                                                              # jittered lattice: near ties everywhere
        m = int(np.ceil(n ** (1 / 3)))
        g = np.stack(np.meshgrid(np.arange(m), np.arange(m), np.arange(m), indexing="ij"), -1).reshape(-1, 3)[rng.permutation(m ** 3)[:n]]
        x = g + rng.normal(0, 1e-3, (n, 3))
    elif kind == 4:                                                              # exact lattice: exact ties at every level
        m = int(np.ceil(n ** (1 / 3)))
        x = np.stack(np.meshgrid(np.arange(m), np.arange(m), np.arange(m), indexing="ij"), -1).reshape(-1, 3)[rng.permutation(m ** 3)[:n]].astype(np.float64)
    elif kind == 5:                                                              # few distinct points
        d = int(rng.integers(5, 400))
        x = rng.normal(0, 5, (d, 3))[rng.integers(0, d, n)]
    elif kind == 6:                                                              # a needle / a sheet
        x = rng.uniform(-1, 1, (n, 3)) * np.array([100.0, 10.0 ** rng.uniform(-4, 0), 10.0 ** rng.uniform(-4, 0)])
    elif kind == 7:                                                              # far from the origin: few bits left for the differences
        x = synthetic.make_scene(n, seed=int(rng.integers(0, 1 << 20))) + rng.uniform(-1, 1, 3) * 10.0 ** rng.uniform(2, 4)
    else:
        x = synthetic.make_scene(n, seed=int(rng.integers(0, 1 << 20))) * 10.0 ** rng.choice([-4.0, 4.0])
    return np.ascontiguousarray(x, dtype=np.float32), kind


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    lib = _lib.load()
    bad = 0
    t0 = time.time()
    for i in range(n_cases):
        n = int(rng.choice([int(rng.integers(1025, 2100)), int(rng.integers(2100, 4097)), int(rng.integers(4097, 8193)), int(rng.integers(8193, 16385))]))
        ratio = float(rng.choice([0.02, 0.1, 0.2, 0.25, 0.5, 1.0], p=[0.1, 0.2, 0.3, 0.2, 0.15, 0.05]))
        x, kind = draw_cloud(rng, n)
        start = int(rng.integers(0, n)) if rng.integers(0, 2) else 0
        ref = G.fps(x, ratio, start=start)
        xd = torch.from_numpy(x).cuda()
        out = torch.full((len(ref),), -7, dtype=torch.int32, device="cuda")
        rc = lib.dedf_fps(xd.data_ptr(), n, len(ref), start, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        got = out.cpu().numpy().astype(np.int64)
        ok = rc == 0 and np.array_equal(got, ref)
        bad += not ok
        print(f"case {i:3d}: kind {kind} n {n:5d} ratio {ratio:4.2f} start {start:5d} samples {len(ref):5d}  "
              f"{'OK' if ok else 'MISMATCH at ' + str(int(np.argmax(got != ref))) + ' rc ' + str(rc)}", flush=True)
    print(f"{n_cases - bad} of {n_cases} cases bit-exact ({time.time() - t0:.0f} s)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
