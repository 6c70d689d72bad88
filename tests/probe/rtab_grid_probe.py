"""Grid of the sampler's radial table: k_edge time, guard words and deviation from the per-edge evaluation for DEDF_RTAB_INF / DEDF_RTAB_FIN (set in the environment)
python tests/probe/rtab_grid_probe.py"""
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
run = lambda: m.sample(Ts, keys, query, [[0.5, 0.5]], [1], [0.04], temperatures=0.0, seed=3)
for _ in range(3): out = run()
st = head.stats()
torch.cuda.synchronize()
head.profile_enable(True); head.profile_read()
for _ in range(20): run()
p = head.profile_read(); head.profile_enable(False)
head.set_radial_table(False)
ref = run()
moved = float((ref[1] - ref[0]).abs().max())
dev_ = float((out[1] - ref[1]).abs().max()) / moved
print(f"INF {os.environ.get('DEDF_RTAB_INF', 'default')} FIN {os.environ.get('DEDF_RTAB_FIN', 'default')}: edge ms {p['ms']['edge'] / p['n_evals']:.4f} (edges {p['n_edges'] / p['n_evals']:.0f}) "
      f"guard {['%.2e' % v for v in st['rtab_err']]} fallback {st['rtab_fallback']} | table vs per-edge step {dev_:.2e} of the displacement")
