"""k_edge / k_aggregate time of the SAMPLER path on C2's seeded poses, ONE step per call (every call sees the same poses, so timing variants whose
kernels write garbage -- e.g. -DDEDF_TIMING_NO_RECORDS -- still do the same edge work): python tests/probe/edge_time_sample_fixed.py  [DEDF_LIB=...]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from diffusion_edf_amd.score_head import ScoreModelHead
from diffusion_edf_amd.score_model_base import ScoreModelBase
dev = torch.device("cuda:0")
NP = int(os.environ.get('POSES', '1000'))          # POSES=16: the small-batch regime (one round of tiles per kernel)
LMAX = int(os.environ.get("LMAX", "2"))
kw, cfg, P, keys, query, Ts = bench.build_inputs(LMAX, 4096, 1024, NP, 0, dev)
head = ScoreModelHead(**kw); head.load_state_dict(P); head.to(dev)
m = ScoreModelBase(head)
def run():
    try:
        m.sample(Ts, keys, query, [[0.5, 0.5]], [1], [1e-9], temperatures=0.0, seed=3)
    except Exception:      # (a garbage-writing timing variant trips the non-finite flag)
        pass
for _ in range(3): run()
torch.cuda.synchronize()
head.profile_enable(True); head.profile_read()
for _ in range(20): run()
torch.cuda.synchronize()
p = head.profile_read()
n = p["n_evals"]
print(os.environ.get("DEDF_LIB", "default"), "edge ms", round(p["ms"]["edge"] / n, 4), "aggregate ms", round(p["ms"]["aggregate"] / n, 4), "node ms", round(p["ms"]["node"] / n, 4), "edges", p["n_edges"] / n)
