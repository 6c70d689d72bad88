"""Which edges differ between the library and the oracle in ONE case of tests/stress_parity.py:  python tests/probe/stress_edges.py <seed> <index>"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import stress_parity as S
import stage_check as SC
import oracle.restatement as R
from diffusion_edf_amd.score_head import ScoreModelHead
from diffusion_edf_amd.gnn_data import FeaturedPoints
seed, idx = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
for i in range(idx):
    S.draw_case(rng)
kw, cfg, P, keys, query, Ts, time = S.draw_case(rng)
ocfg = R.config_from_kwargs(kw)._replace(max_neighbors=cfg.max_neighbors)
ok = [R.FeaturedPoints(k.x.double(), k.f.double(), k.b, None) for k in keys]
oq = R.FeaturedPoints(query.x.double(), query.f.double(), query.b, query.w.double())
dbg = R.Debug()
R.score_head_forward(ocfg, R.cast_params(P, torch.float64), Ts, ok, oq, time, dbg)
dev = torch.device('cuda:0')
head = ScoreModelHead(**kw); head.cfg.max_neighbors = cfg.max_neighbors
head.load_state_dict(P); head.to(dev)
gk = [FeaturedPoints(k.x.to(dev), k.f.to(dev), k.b.to(dev), None) for k in keys]
gq = FeaturedPoints(query.x.to(dev), query.f.to(dev), query.b.to(dev), query.w.to(dev))
head.set_key_clouds(gk); head.set_query(gq); head.debug_enable(True)
head(Ts.to(dev, torch.float32), gk, gq, time.to(dev, torch.float32)); torch.cuda.synchronize()
es = head.debug_buffer('edge_src', torch.int32).long(); ed = head.debug_buffer('edge_dst', torch.int32).long()
ti = head.debug_buffer('tile_info', torch.int32)
E = int(ti[16 + cfg.n_scales]); es, ed = es[:E], ed[:E]
nkeys = sum(len(k.x) for k in keys)
g = set((int(d) * nkeys + int(s)) for s, d in zip(es.tolist(), ed.tolist()))
o = set((int(d) * nkeys + int(s)) for s, d in zip(dbg['edge_src'].tolist(), dbg['edge_dst'].tolist()))
only_g, only_o = sorted(g - o), sorted(o - g)
print("E", E, "only gpu", len(only_g), "only oracle", len(only_o), "cap", cfg.max_neighbors, "keys", [len(k.x) for k in keys])
xq = R.transform_points(query.x.double(), Ts).reshape(-1, 3)
allx = torch.cat([k.x.double() for k in keys])
starts = np.cumsum([0] + [len(k.x) for k in keys])
for key in (only_g[:4] + only_o[:4]):
    d, s = key // nkeys, key % nkeys
    n = int(np.searchsorted(starts, s, side='right') - 1)
    dist = float((xq[d] - allx[s]).norm())
    gl = sorted(int(k % nkeys) for k in g if k // nkeys == d and starts[n] <= k % nkeys < starts[n + 1])
    ol = sorted(int(k % nkeys) for k in o if k // nkeys == d and starts[n] <= k % nkeys < starts[n + 1])
    cand = [(int(j + starts[n]), float((xq[d] - keys[n].x[j].double()).norm())) for j in range(len(keys[n].x)) if float((xq[d] - keys[n].x[j].double()).norm()) < cfg.radii[n] * 1.0001]
    print("dst", d, "scale", n, "src", s, "dist", dist, "r", cfg.radii[n], "in", "gpu" if key in g else "oracle")
    print("   gpu   :", gl)
    print("   oracle:", ol)
    print("   candidates (index, dist):", [(c, round(x, 6)) for c, x in cand][:30])
