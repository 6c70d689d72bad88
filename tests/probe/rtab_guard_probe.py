import sys, torch, numpy as np
sys.path.insert(0, "tests")
import stage_check as SC
from diffusion_edf_amd.gnn_data import FeaturedPoints
from diffusion_edf_amd.score_head import ScoreModelHead
from diffusion_edf_amd.score_model_base import ScoreModelBase
dev = torch.device('cuda:0')
for lmax in (2, 3):
    kw, cfg, P, keys, query, Ts, _ = SC.build_case(lmax, 48, 2048, 256)
    for tag, sig in (("init", None), ("3e-3", 3e-3), ("1e-3", 1e-3)):
        Q = {k: v.clone() for k, v in P.items()}
        if sig:
            for n in (0, 2):
                Q[f"key_tensor_field.graph_parsers.{n}.length_enc.param_module.std_logit"][:] = float(np.log(np.expm1(sig)))
        head = ScoreModelHead(**kw); head.load_state_dict(Q); head.to(dev)
        gk = [FeaturedPoints(k.x.to(dev), k.f.to(dev), k.b.to(dev)) for k in keys]
        gq = FeaturedPoints(query.x.to(dev), query.f.to(dev), query.b.to(dev), query.w.to(dev))
        for t in (0.9, 0.05):
            head.set_radial_table("always")
            o = ScoreModelBase(head).sample(Ts.to(dev), gk, gq, [[t, t]], [1], [0.04], temperatures=0.0).cpu()
            st = head.stats()
            head.set_radial_table(False)
            o2 = ScoreModelBase(head).sample(Ts.to(dev), gk, gq, [[t, t]], [1], [0.04], temperatures=0.0).cpu()
            d1, d2 = (o[1]-o[0])[:,4:], (o2[1]-o2[0])[:,4:]
            print(lmax, tag, t, ["%.2e" % e for e in st['rtab_err']], st['rtab_fallback'], "on-vs-off %.2e" % (float((d1-d2).abs().max())/float(d2.abs().max())))
