"""The reference's deployment shapes (round-5 review, item 3): 20 poses x the 2 static keypoints of pick_lowres (configs/panda_mug/pick_lowres/score_model_configs.yaml:76-80)
and 20 poses x 103 query points, in 200-step calls (configs/panda_mug/server.yaml:2): wall time per step and kernel time per class.
    python tests/probe/deploy_shape.py [steps]          (DEDF_RADIAL_TABLE / DEDF_EDGE16 / DEDF_SMALL_BATCH are read by the library)"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from diffusion_edf_amd.gnn_data import FeaturedPoints
from diffusion_edf_amd.score_head import ScoreModelHead
from diffusion_edf_amd.score_model_base import ScoreModelBase

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda:0")
kw, cfg, P, keys, query, Ts = bench.build_inputs(2, 4096, 1024, 20, 0, dev)
g = torch.Generator().manual_seed(5)
q2 = FeaturedPoints(x=torch.tensor([[0.5, 0.5, 10.5], [-0.5, -0.5, 10.5]], device=dev), f=torch.randn(2, query.f.shape[1], generator=g).to(dev),
                    b=torch.zeros(2, dtype=torch.long, device=dev), w=torch.sigmoid(torch.randn(2, generator=g)).to(dev))
head = ScoreModelHead(**kw); head.load_state_dict(P); head.to(dev)
m = ScoreModelBase(head)
for label, q in (("20 poses x 2 static keypoints", q2), ("20 poses x 103 query points", query)):
    run = lambda n: m.sample(Ts, keys, q, [[1.0, 0.15]], [n], [0.04], temperatures=1.0, seed=3)
    run(5)
    torch.cuda.synchronize()
    reps = []
    for _ in range(5):
        t0 = time.perf_counter(); run(steps); torch.cuda.synchronize(); reps.append((time.perf_counter() - t0) / steps * 1e3)
    reps.sort()
    head.profile_enable(True); head.profile_read()
    run(steps)
    pr = head.profile_read(); head.profile_enable(False)
    us = {k: round(v / max(1, pr["n_evals"]) * 1e3, 1) for k, v in pr["ms"].items()}
    print(f"{label}: median {reps[2]:.4f} ms/step (min {reps[0]:.4f}, max {reps[-1]:.4f}), edges/step {pr['n_edges'] / max(1, pr['n_evals']):.0f}, kernel us/step {us} sum {sum(us.values()):.1f}", flush=True)
