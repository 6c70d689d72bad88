"""Probe: error of the lmax-3 UNet feature extractor on the 16 384-point scene, per output scale and irreps block, against the fp64
restatement -- the HIP library in fp32 mode, in fp16-GEMM mode, and the restatement itself run in fp32 (the floor a 17-layer fp32 chain has).
Usage: python tests/probe/unet_err_lmax3.py [n_points]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from diffusion_edf_amd import synthetic
from diffusion_edf_amd.gnn_data import FeaturedPoints
from diffusion_edf_amd.so3 import parse_irreps
from diffusion_edf_amd.unet import UnetFeatureExtractor
from oracle import restatement as R, unet_oracle as U
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from test_lmax3 import _randomized, IRREPS3, SH3

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
dev = torch.device("cuda:0")
m = UnetFeatureExtractor(**synthetic.unet_kwargs("panda_lowres_lmax3"), deterministic=True)
sd = _randomized(m, seed=5)
kw = m._ctor
ocfg = U.UnetConfig(irreps_input=parse_irreps(kw["irreps_input"]), irreps_output=parse_irreps(kw["irreps_output"]),
                    irreps_emb=[parse_irreps(i) for i in kw["irreps_emb"]], fc_neurons=[list(f) for f in kw["fc_neurons"]],
                    n_layers=list(kw["n_layers"]), pool_ratio=list(kw["pool_ratio"]), radius=list(m.radius),
                    n_layers_midstream=kw["n_layers_midstream"], irreps_sh=SH3)
x = torch.from_numpy(synthetic.make_scene(n, seed=0).astype(np.float32))
f = torch.rand(n, 3, generator=torch.Generator().manual_seed(1))
t0 = time.time()
ref = U.unet_forward(ocfg, R.cast_params(sd, torch.float64), x, f.double())
print(f"fp64 restatement: {time.time() - t0:.1f} s", flush=True)

def report(name, outs):
    for (xr, fr), got in zip(ref, outs):
        off, row = 0, []
        for mul, l in IRREPS3:
            d = mul * (2 * l + 1)
            row.append(float((got[:, off:off + d].double() - fr[:, off:off + d]).abs().max()) / max(float(fr[:, off:off + d].abs().max()), 1e-3 * float(fr.abs().max())))
            off += d
        print(f"{name:>22s}  scale {len(xr):5d}: " + "  ".join(f"l{l}={e:.2e}" for (_, l), e in zip(IRREPS3, row)), flush=True)

try:
    t0 = time.time()
    r32 = U.unet_forward(ocfg, R.cast_params(sd, torch.float32), x, f.float())
    print(f"fp32 restatement: {time.time() - t0:.1f} s")
    report("restatement in fp32", [fr for _, fr in r32])
except Exception as e:      # the restatement may insist on fp64 somewhere
    print("fp32 restatement failed:", repr(e))
m.to(dev)
fp = FeaturedPoints(x=x.to(dev), f=f.to(dev), b=torch.zeros(n, dtype=torch.long, device=dev), w=None)
report("HIP fp32 (3-term)", [o.f.cpu() for o in m(fp)])
m.half()
report("HIP fp16-GEMM mode", [o.f.cpu() for o in m(fp)])
