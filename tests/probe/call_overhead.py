"""Per-call overhead of ScoreModelBase.sample: wall time of calls with 1 / 10 / 50 / 200 steps -> fit  t = overhead + steps * per_step
python tests/probe/call_overhead.py"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from diffusion_edf_amd.score_head import ScoreModelHead
from diffusion_edf_amd.score_model_base import ScoreModelBase
dev = torch.device("cuda:0")
for lm, scene, grasp, nT in ((1, 2048, 512, 256), (2, 4096, 1024, 16), (2, 4096, 1024, 1000)):
    kw, cfg, P, keys, query, Ts = bench.build_inputs(lm, scene, grasp, nT, 0, dev)
    head = ScoreModelHead(**kw); head.load_state_dict(P); head.to(dev)
    m = ScoreModelBase(head)
    run = lambda n: m.sample(Ts, keys, query, [[1.0, 0.15]], [n], [0.04], temperatures=1.0, seed=3)
    run(5); torch.cuda.synchronize()
    rows = []
    for n in (1, 10, 50, 200):
        ts = []
        for _ in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter(); run(n); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        rows.append((n, sorted(ts)[2] * 1e3))
    A = np.array([[1.0, r[0]] for r in rows]); b = np.array([r[1] for r in rows])
    (ov, per), *_ = np.linalg.lstsq(A, b, rcond=None)
    print(f"lmax {lm} nT {nT}: " + ", ".join(f"{n} steps {t:.3f} ms" for n, t in rows) + f"  -> overhead {ov:.3f} ms + {per:.4f} ms/step")
