"""Replay one case of tests/stress_parity.py (python tests/probe/replay_case.py <seed> <index>) and compare edge SETS and scores
with the fp64 and the fp32 restatement."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import stress_parity as SP
import oracle.restatement as R
from diffusion_edf_amd.score_head import ScoreModelHead
from diffusion_edf_amd.gnn_data import FeaturedPoints
seed, idx = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
for i in range(idx + 1):
    case = SP.draw_case(rng)
kw, cfg, P, keys, query, Ts, time = case
ocfg = R.config_from_kwargs(kw)._replace(max_neighbors=cfg.max_neighbors)
res = {}
for name, dt in (("f64", torch.float64), ("f32", torch.float32)):
    ok = [R.FeaturedPoints(k.x.to(dt), k.f.to(dt), k.b, None if k.w is None else k.w.to(dt)) for k in keys]
    oq = R.FeaturedPoints(query.x.to(dt), query.f.to(dt), query.b, query.w.to(dt))
    d = R.Debug()
    ang, lin = R.score_head_forward(ocfg, R.cast_params(P, dt), Ts.to(dt), ok, oq, time.to(dt), d)
    res[name] = (ang.double(), lin.double(), set(zip(d['edge_dst'].tolist(), d['edge_src'].tolist())))
dev = torch.device('cuda:0')
head = ScoreModelHead(**kw); head.cfg.max_neighbors = cfg.max_neighbors; head.load_state_dict(P); head.to(dev)
gk = [FeaturedPoints(k.x.to(dev), k.f.to(dev), k.b.to(dev), None if k.w is None else k.w.to(dev)) for k in keys]
gq = FeaturedPoints(query.x.to(dev), query.f.to(dev), query.b.to(dev), query.w.to(dev))
head.debug_enable(True)
ang, lin = head(Ts.to(dev, torch.float32), gk, gq, time.to(dev, torch.float32))
E = head.stats()['n_edges_total']
es = head.debug_buffer('edge_src', torch.int32)[:E].tolist(); ed = head.debug_buffer('edge_dst', torch.int32)[:E].tolist()
gset = set(zip(ed, es))
for name in ("f64", "f32"):
    a, l, s = res[name]
    scale = float(max(a.abs().max(), l.abs().max()))
    err = max(float((ang.cpu().double() - a).abs().max()), float((lin.cpu().double() - l).abs().max())) / scale
    print(name, "err", err, "edges only in gpu", len(gset - s), "only in oracle", len(s - gset))
a64, l64, _ = res["f64"]; a32, l32, _ = res["f32"]
print("f32 oracle vs f64 oracle", max(float((a32 - a64).abs().max()), float((l32 - l64).abs().max())) / float(max(a64.abs().max(), l64.abs().max())))
