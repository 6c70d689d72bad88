"""radial table on/off vs the fp64 oracle: one noise-free sampler step.  python tests/probe/rtab_dev.py"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import stage_check as SC
from diffusion_edf_amd.gnn_data import FeaturedPoints
from diffusion_edf_amd.score_head import ScoreModelHead
from diffusion_edf_amd.score_model_base import ScoreModelBase
from oracle import restatement as R
dev = torch.device("cuda:0")
for radii in ((5., 10., 20., None), (3.5, 5., 6.5, 8.), (None,), (5.,)):
    kw, cfg, P, keys, query, Ts, time = SC.build_case(2, 12, 1024, 128, radii=radii)
    ocfg = R.config_from_kwargs(kw)
    head = ScoreModelHead(**kw); head.load_state_dict(P); head.to(dev)
    gk = [FeaturedPoints(k.x.to(dev), k.f.to(dev), k.b.to(dev)) for k in keys]
    gq = FeaturedPoints(query.x.to(dev), query.f.to(dev), query.b.to(dev), query.w.to(dev))
    P64 = R.cast_params(P, torch.float64)
    k64 = [R.FeaturedPoints(k.x.double(), k.f.double(), k.b, None) for k in keys]
    q64 = R.FeaturedPoints(query.x.double(), query.f.double(), query.b, query.w.double())
    for t in (0.9, 0.05):
        ang, lin = R.score_head_forward(ocfg, P64, Ts, k64, q64, torch.full((len(Ts),), t, dtype=torch.float64))
        z = torch.zeros(len(Ts), 3, dtype=torch.float64)
        ref = R.langevin_step(ocfg, Ts, ang, lin, t, 0.04, 0.0, 0.5, 0.5, z, z)
        outs = []
        for on in ("always", False):
            head.set_radial_table(on)
            outs.append(ScoreModelBase(head).sample(Ts.to(dev), gk, gq, [[t, t]], [1], [0.04], temperatures=0.0).cpu())
        d_ref = (ref - Ts)[:, 4:]
        d_on, d_off = (outs[0][1] - outs[0][0])[:, 4:], (outs[1][1] - outs[1][0])[:, 4:]
        s = float(d_ref.abs().max())
        print(f"radii {radii} t {t}: on-off {float((d_on - d_off).abs().max()) / s:.2e}  on-oracle {float((d_on - d_ref).abs().max()) / s:.2e}  off-oracle {float((d_off - d_ref).abs().max()) / s:.2e}")
