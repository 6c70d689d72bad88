"""Small-batch sampler step (the deployment regime: 10-20 poses x 650-900 steps, reference evaluate_real_mug.ipynb:188-190, configs/panda_mug/server.yaml:2):
wall time per step and summed kernel time per class.   python tests/probe/small_batch.py [lmax] [steps]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from diffusion_edf_amd.score_head import ScoreModelHead
from diffusion_edf_amd.score_model_base import ScoreModelBase

lmax = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda:0")
for scene, grasp, nT in ((4096, 1024, 16), (4096, 1024, 64), (4096, 1024, 256), (2048, 512, 256)):
    lm = 1 if scene == 2048 else lmax
    kw, cfg, P, keys, query, Ts = bench.build_inputs(lm, scene, grasp, nT, 0, dev)
    head = ScoreModelHead(**kw); head.load_state_dict(P); head.to(dev)
    m = ScoreModelBase(head)
    run = lambda n: m.sample(Ts, keys, query, [[1.0, 0.15]], [n], [0.04], temperatures=1.0, seed=3)
    run(5)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); run(steps); torch.cuda.synchronize(); el = time.perf_counter() - t0
    head.profile_enable(True); head.profile_read()
    run(steps)
    pr = head.profile_read(); head.profile_enable(False)
    ms = {k: round(v / max(1, pr["n_evals"]) * 1e3, 1) for k, v in pr["ms"].items()}
    print(f"lmax {lm} scene {scene} nT {nT}: {el / steps * 1e3:.4f} ms/step, {nT * steps / el:.0f} pose-steps/s, edges/step {pr['n_edges'] / max(1, pr['n_evals']):.0f}, kernel us/step {ms} sum {sum(ms.values()):.1f}")
