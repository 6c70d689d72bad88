"""timing experiment: one 16-pose sampler call (wrong results allowed: DEDF_LIB may point at a timing build of the fill pass)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from diffusion_edf_amd.score_head import ScoreModelHead
from diffusion_edf_amd.score_model_base import ScoreModelBase
dev = torch.device("cuda:0")
nT = int(sys.argv[1]) if len(sys.argv) > 1 else 16
kw, cfg, P, keys, query, Ts = bench.build_inputs(2, 4096, 1024, nT, 0, dev)
head = ScoreModelHead(**kw); head.load_state_dict(P); head.to(dev)
m = ScoreModelBase(head)
try:
    m.sample(Ts, keys, query, [[1.0, 0.15]], [100], [0.04], temperatures=1.0, seed=3)
except Exception as e:
    print("sample raised:", str(e)[:100])
torch.cuda.synchronize()
