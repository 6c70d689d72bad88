"""k_edge time on FIXED poses (C2 inputs of bench.py, t = 0.5), for timing probes whose kernels compute garbage."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from diffusion_edf_amd.score_head import ScoreModelHead

dev = torch.device("cuda:0")
kw, cfg, P, keys, query, Ts = bench.build_inputs(2, 4096, 1024, 1000, 0, dev)
head = ScoreModelHead(**kw); head.load_state_dict(P); head.to(dev)
head.set_key_clouds(keys); head.set_query(query)
T = Ts.float(); t = torch.full((len(T),), 0.5, device=dev)
for _ in range(3): head(T, keys, query, t)
torch.cuda.synchronize()
head.profile_enable(True); head.profile_read()
for _ in range(10): head(T, keys, query, t)
torch.cuda.synchronize()
p = head.profile_read()
print(os.environ.get("DEDF_LIB", "default"), "edge ms", p["ms"]["edge"] / p["n_evals"], "edges", p["n_edges"] / p["n_evals"])
