"""k_edge time of the SAMPLER path (shared time: radial table) on C2 inputs: python tests/probe/edge_time_sample.py   [DEDF_LIB=...]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from diffusion_edf_amd.score_head import ScoreModelHead
from diffusion_edf_amd.score_model_base import ScoreModelBase
dev = torch.device("cuda:0")
kw, cfg, P, keys, query, Ts = bench.build_inputs(2, 4096, 1024, 1000, 0, dev)
head = ScoreModelHead(**kw); head.load_state_dict(P); head.to(dev)
m = ScoreModelBase(head)
run = lambda n: m.sample(Ts, keys, query, [[1.0, 0.15]], [n], [0.04], temperatures=1.0, seed=3)
run(3); torch.cuda.synchronize()
head.profile_enable(True); head.profile_read()
run(20); torch.cuda.synchronize()
p = head.profile_read()
print(os.environ.get("DEDF_LIB", "default"), "edge ms", p["ms"]["edge"] / p["n_evals"], "edges", p["n_edges"] / p["n_evals"])
