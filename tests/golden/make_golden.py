"""Generate tests/golden/*.npz by importing the two reference modules that are importable in the build container:
/root/reference/diffusion_edf/transforms.py (pure torch) and /root/reference/diffusion_edf/radial_func.py (with a
no-op `beartype` shim).  Run ONLY in the build container:   python tests/golden/make_golden.py
The .npz files hold inputs and the reference's outputs (data, not source); nothing from /root/reference travels.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/diffusion_edf"

# minimal no-op beartype shim so radial_func.py imports
bt = types.ModuleType("beartype")
bt.beartype = lambda f: f
sys.modules["beartype"] = bt


def load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


T = load("ref_transforms", os.path.join(REF, "transforms.py"))
RF = load("ref_radial_func", os.path.join(REF, "radial_func.py"))

g = torch.Generator().manual_seed(0)
out = {}

# ---- transforms.py -----------------------------------------------------------------------------------------------------
q = torch.randn(64, 4, generator=g, dtype=torch.float64)
q = q / q.norm(dim=-1, keepdim=True)
special = torch.tensor([[1., 0, 0, 0], [0., 1, 0, 0], [0., 0, 1, 0], [0., 0, 0, 1], [-1., 0, 0, 0],
                        [0.5, 0.5, 0.5, 0.5], [2 ** -0.5, 0, 2 ** -0.5, 0]], dtype=torch.float64)
q = torch.cat([special, q], 0)
p = torch.randn(len(q), 3, generator=g, dtype=torch.float64)
out["q"] = q.numpy()
out["p"] = p.numpy()
out["standardize"] = T.standardize_quaternion(q).numpy()
out["to_matrix"] = T.quaternion_to_matrix(q).numpy()
out["euler_yxy_f64"] = T.matrix_to_euler_angles(T.quaternion_to_matrix(T.standardize_quaternion(q)), "YXY").numpy()
q32 = q.float()
out["euler_yxy_f32"] = T.matrix_to_euler_angles(T.quaternion_to_matrix(T.standardize_quaternion(q32)), "YXY").numpy()
out["apply"] = T.quaternion_apply(q, p).numpy()
out["invert"] = T.quaternion_invert(q).numpy()
out["raw_multiply"] = T.quaternion_raw_multiply(q, q.flip(0)).numpy()
out["normalize"] = T.normalize_quaternion(q * 3.0).numpy()
np.savez(os.path.join(HERE, "transforms.npz"), **out)

# ---- radial_func.py ------------------------------------------------------------------------------------------------------
out = {}
x = torch.linspace(0, 6, 13)
out["ssc2_x"] = x.numpy()
out["ssc2_right"] = RF.soft_square_cutoff_2(x, (None, None, 4., 5.)).numpy()
x2 = torch.linspace(0, 0.5, 101)
out["ssc2_x2"] = x2.numpy()
out["ssc2_left"] = RF.soft_square_cutoff_2(x2, (0.2 * 0.3, 0.3, None, None)).numpy()
x3 = torch.linspace(0, 25, 251)
out["ssc2_x3"] = x3.numpy()
out["ssc2_r20"] = RF.soft_square_cutoff_2(x3, (None, None, 0.8 * 20., 20.)).numpy()
out["soft_step"] = RF.soft_step(torch.linspace(-0.5, 1.5, 41)).numpy()
d = torch.rand(200, generator=g) * 22.0
out["dist"] = d.numpy()
for r in (5., 10., 20.):
    enc = RF.GaussianRadialBasis(dim=64, max_val=r)
    enc.eval()
    with torch.no_grad():
        out[f"grb_{int(r)}"] = enc(d).numpy()
sin = RF.SinusoidalPositionEmbeddings(dim=64, max_val=100., n=1000.)
out["sinus_len"] = sin(d * 3.0).numpy()
tt = torch.rand(50, generator=g)
out["time"] = tt.numpy()
out["sinus_time"] = RF.SinusoidalPositionEmbeddings(dim=256, max_val=1., n=10000.)(tt).numpy()
out["sinus_time_f64"] = RF.SinusoidalPositionEmbeddings(dim=256, max_val=1., n=10000.)(tt.double()).numpy()
np.savez(os.path.join(HERE, "radial_func.npz"), **out)

# ---- radial_func.py: GaussianRadialBasisLayerFiniteCutoff (the radial basis of the UNet blocks, unet_feature_extractor.py:144) -----
out = {}
g2 = torch.Generator().manual_seed(7)
d = torch.cat([torch.rand(300, generator=g2) * 7.5, torch.tensor([0.0, 0.01, 0.066, 0.0663, 0.07, 1.3, 6.63, 6.7])])
out["dist"] = d.numpy()
for nb, r in ((64, 6.7), (64, 15.0), (32, 3.0)):
    lay = RF.GaussianRadialBasisLayerFiniteCutoff(num_basis=nb, cutoff=0.99 * r)
    with torch.no_grad():      # trained-looking parameters, so that every term of the formula is pinned
        lay.mean.add_(torch.randn(1, nb, generator=g2) * 0.02)
        lay.std_logit.add_(torch.randn(1, nb, generator=g2) * 0.3)
        lay.weight_logit.add_(torch.randn(1, nb, generator=g2) * 0.5)
        tag = f"{nb}_{str(r).replace('.', 'p')}"
        out[f"mean_{tag}"] = lay.mean.numpy().copy()
        out[f"std_logit_{tag}"] = lay.std_logit.numpy().copy()
        out[f"weight_logit_{tag}"] = lay.weight_logit.numpy().copy()
        out[f"cutoff_offset_{tag}"] = np.array([lay.cutoff, lay.offset])
        out[f"out_{tag}"] = lay(d).numpy()
        out[f"out_f64_{tag}"] = lay.double()(d.double()).numpy()
np.savez(os.path.join(HERE, "unet_radial.npz"), **out)
print("wrote", os.listdir(HERE))
