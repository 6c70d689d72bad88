# Per-phase cycle profile of k_edge.  Build the instrumented library first (single translation unit):
#   cd diffusion_edf_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++20 -shared -fPIC -DDEDF_SINGLE_TU -DDEDF_PHASE_PROF -I../../include dedf_api.hip -o libdedf_prof.so
#   DEDF_LIB=diffusion_edf_amd/csrc/libdedf_prof.so python tests/phase_prof.py
import os, sys, torch, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from diffusion_edf_amd.score_head import ScoreModelHead
dev = torch.device('cuda:0')
LMAX = int(os.environ.get('LMAX', '2'))          # LMAX=3: build unit 22 (sampler) / 16 (score forward) with -DDEDF_PHASE_PROF
kw, cfg, P, keys, query, Ts = bench.build_inputs(LMAX, 4096, 1024, 1000, 0, dev)
head = ScoreModelHead(**kw); head.load_state_dict(P); head.to(dev)
t = torch.full((1000,), 0.5, device=dev)
if os.environ.get("SAMPLE"):          # the sampler's path (shared time: radial table) instead of the score forward
    from diffusion_edf_amd.score_model_base import ScoreModelBase
    _m = ScoreModelBase(head)
    run = lambda: _m.sample(Ts, keys, query, [[0.5, 0.5]], [1], [1e-9], temperatures=0.0)
else:
    run = lambda: head(Ts.float(), keys, query, t)
for _ in range(3): run()
torch.cuda.synchronize()
# zero prof
import ctypes as C
from diffusion_edf_amd import _lib
buf = head.debug_buffer('phase_prof')   # float32 view of u64 data
raw0 = buf.numpy().view(np.uint64).reshape(-1, 16).copy()
run(); torch.cuda.synchronize()
raw1 = head.debug_buffer('phase_prof').numpy().view(np.uint64).reshape(-1, 16)
d = (raw1 - raw0).astype(np.float64)
names = ["geom+enc", "prelin+silu", "L1 mfma", "L1 LN+silu", "L2 mfma", "L2 LN+silu", "accinit", "E prologue(wt0)", "E l3=0 chunks", "E l3=1 chunks", "F l3=0 chunks", "stores+end", "E l3=2 chunks" if LMAX == 2 else "E l3=3 chunks", "F l3=1 chunks", "F l3=2 chunks" if LMAX == 2 else "F l3=3 chunks", "-" if LMAX == 2 else "E l3=2 chunks"]
if LMAX == 3: names[2] = "F l3=2 chunks"
tot = d[:, :16].sum(1).mean()
E = head.stats()['n_edges_total']; tiles = sum((e + 31)//32 for e in head.stats()['n_edges'])
print("edges", E, "tiles", tiles, "tiles/wave", tiles / d.shape[0])
for i, n in enumerate(names):
    print(f"{n:18s} {d[:, i].mean() / (tiles / d.shape[0]):10.0f} cycles/tile  {100 * d[:, i].mean() / tot:5.1f}%")
# (SAMPLE=1, table-reading kernel: slots 0 / 1 = tile start -> geometry done -> table rows requested; 2-4 unused; 5 = combine; 6 = up to the fused stage)
print("total cycles/tile", tot / (tiles / d.shape[0]))
