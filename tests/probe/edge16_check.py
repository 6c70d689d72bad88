"""The 16-edge / two-waves-per-SIMD edge kernel (DEDF_EDGE16=1, dedf_edge16.h) against the 32-edge kernel and the fp64 oracle: one noise-free
sampler step (displacement = alpha / 2 * score), then timing on C2.   python tests/probe/edge16_check.py [time_only]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import stage_check as SC, bench
from diffusion_edf_amd.gnn_data import FeaturedPoints
from diffusion_edf_amd.score_head import ScoreModelHead
from diffusion_edf_amd.score_model_base import ScoreModelBase
from oracle import restatement as R
dev = torch.device("cuda:0")

def head_for(kw, P, e16):
    os.environ["DEDF_EDGE16"] = str(e16)
    h = ScoreModelHead(**kw); h.load_state_dict(P); h.to(dev); h.set_radial_table("always")
    return h

STEP = float(os.environ.get("STEP", "0.04"))          # STEP=1e-12: poses (hence edges) stay put -- for the wrong-result timing variants
if len(sys.argv) < 2:
    for radii, nT in (((5., 10., 20., None), 12), ((3.5, 5., 6.5, 8.), 40)):
        kw, cfg, P, keys, query, Ts, _ = SC.build_case(2, nT, 1024, 128, radii=radii)
        gk = [FeaturedPoints(k.x.to(dev), k.f.to(dev), k.b.to(dev)) for k in keys]
        gq = FeaturedPoints(query.x.to(dev), query.f.to(dev), query.b.to(dev), query.w.to(dev))
        t = 0.5
        outs = {}
        for e16 in (0, 1):
            h = head_for(kw, P, e16)
            outs[e16] = ScoreModelBase(h).sample(Ts.to(dev), gk, gq, [[t, t]], [1], [0.04], temperatures=0.0).cpu()
            print("edges", h.stats()["n_edges"])
        d0, d1 = outs[0][1] - outs[0][0], outs[1][1] - outs[1][0]
        sc_q, sc_x = float(d0[:, :4].abs().max()), float(d0[:, 4:].abs().max())
        print(radii, "16-edge vs 32-edge: rot %.2e lin %.2e" % (float((d1 - d0)[:, :4].abs().max()) / sc_q, float((d1 - d0)[:, 4:].abs().max()) / sc_x))
        ocfg = R.config_from_kwargs(kw)
        k64 = [R.FeaturedPoints(k.x.double(), k.f.double(), k.b, None) for k in keys]
        q64 = R.FeaturedPoints(query.x.double(), query.f.double(), query.b, query.w.double())
        ang, lin = R.score_head_forward(ocfg, R.cast_params(P, torch.float64), Ts, k64, q64, torch.full((len(Ts),), t, dtype=torch.float64))
        z = torch.zeros(len(Ts), 3, dtype=torch.float64)
        dr = R.langevin_step(ocfg, Ts, ang, lin, t, 0.04, 0.0, 0.5, 0.5, z, z) - Ts
        for e16 in (0, 1):
            d = outs[e16][1] - outs[e16][0]
            print("  EDGE16=%d vs fp64 oracle: rot %.2e lin %.2e" % (e16, float((d - dr)[:, :4].abs().max()) / sc_q, float((d - dr)[:, 4:].abs().max()) / sc_x))
# timing on C2
kw, cfg, P, keys, query, Ts = bench.build_inputs(2, 4096, 1024, 1000, 0, dev)
for e16 in (0, 1, 0, 1):
    h = head_for(kw, P, e16)
    m = ScoreModelBase(h)
    run = lambda n: m.sample(Ts, keys, query, [[1.0, 0.15]], [n], [STEP], temperatures=1.0 if STEP > 1e-6 else 0.0, seed=3)
    run(3); torch.cuda.synchronize()
    h.profile_enable(True); h.profile_read()
    t0 = time.perf_counter(); run(20); torch.cuda.synchronize(); el = time.perf_counter() - t0
    pr = h.profile_read()
    print("C2 EDGE16=%d: %.3f ms/step, %.0f pose-steps/s, k_edge %.3f ms at %.0f edges, aggregate %.3f" % (e16, el / 20 * 1e3, 1000 * 20 / el, pr["ms"]["edge"] / pr["n_evals"], pr["n_edges"] / pr["n_evals"], pr["ms"]["aggregate"] / pr["n_evals"]))
