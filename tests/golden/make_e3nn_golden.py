"""The e3nn convention pin, as a switch ("parity unpinned" because e3nn 0.4.4 is not importable in the build container).

Run this where `e3nn==0.4.4` IS importable (and, for the module-level sections, the reference checkout with its dependencies):

    python tests/golden/make_e3nn_golden.py [/path/to/reference]        ->  tests/golden/e3nn_0_4_4.npz

It dumps inputs and outputs of e3nn and of the reference's own modules (data, not source):

  w3j_<l1><l2><l3>      o3.wigner_3j for every triple l <= 3                        (reference call sites: equiformer/tensor_product_rescale.py:38-42)
  sh_points, sh_<l>     o3.spherical_harmonics(l, x, normalize=True, 'component')   (graph_parser.py:135)
  n2m_*                 e3nn.math.normalize2mom constants of SiLU / sigmoid / SmoothLeakyReLU(0.2)   (fast_activation.py:69)
  tp_*                  o3.TensorProduct 'uvu' / 'uvw' with path_normalization='none' on small irreps, external weights (flat layout)
  sfctp_*               one SeparableFCTP forward of the reference (graph_attention_transformer.py:60-135): state dict, inputs, output
  block_*               one EquiformerBlock forward of the reference (gnn_block.py:60-218): state dict, inputs, output

Every section is optional (try / except): whatever could be produced is listed in `sections`.  tests/test_e3nn_pin.py compares
oracle/so3_oracle.py / oracle/restatement.py (and the product's generated tables) with whatever the file holds, and is SKIPPED while the file is
absent -- committing the file flips SURVEY 8(c) to "pinned" with no other change."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
out, sections = {}, []

try:
    import e3nn
    from e3nn import o3
except Exception as e:          # noqa: BLE001
    raise SystemExit(f"e3nn is not importable here ({e}); run this script in an environment with e3nn==0.4.4")
out["e3nn_version"] = np.array(e3nn.__version__)
if e3nn.__version__ != "0.4.4":
    print(f"warning: e3nn {e3nn.__version__}, the reference pins 0.4.4 (setup.py:28)")
torch.set_default_dtype(torch.float64)
g = torch.Generator().manual_seed(0)

# ---- 3j symbols ----------------------------------------------------------------------------------------------------------------------------
for l1 in range(4):
    for l2 in range(4):
        for l3 in range(abs(l1 - l2), min(3, l1 + l2) + 1):
            out[f"w3j_{l1}{l2}{l3}"] = o3.wigner_3j(l1, l2, l3).numpy()
sections.append("w3j")

# ---- spherical harmonics -------------------------------------------------------------------------------------------------------------------
pts = torch.cat([torch.eye(3), -torch.eye(3), torch.tensor([[1.0, 1.0, 0.0], [0.0, 1.0, 1.0], [1.0, 0.0, 1.0], [1.0, 2.0, 3.0]]),
                 torch.randn(54, 3, generator=g)], 0)
out["sh_points"] = pts.numpy()
for l in range(4):
    out[f"sh_{l}"] = o3.spherical_harmonics(l, pts, normalize=True, normalization='component').numpy()
sections.append("sh")

# ---- normalize2mom ---------------------------------------------------------------------------------------------------------------------------
try:
    from e3nn.math import normalize2mom

    class _SLReLU(torch.nn.Module):          # fast_activation.py:14-23
        def __init__(self, a=0.2):
            super().__init__()
            self.a = a

        def forward(self, x):
            return ((1 + self.a) / 2) * x + ((1 - self.a) / 2) * x * (2 * torch.sigmoid(x) - 1)

    for name, f in (("silu", torch.nn.SiLU()), ("sigmoid", torch.sigmoid), ("slrelu", _SLReLU(0.2))):
        out[f"n2m_{name}"] = np.array(float(normalize2mom(f).cst))
    sections.append("n2m")
except Exception as e:          # noqa: BLE001
    print("normalize2mom section skipped:", e)

# ---- o3.TensorProduct, path_normalization='none' (what TensorProductRescale builds, tensor_product_rescale.py:38-42) ------------------------------
try:
    ir1, ir2 = o3.Irreps("4x0e+3x1e+2x2e"), o3.Irreps("1x0e+1x1e+1x2e")
    iro = o3.Irreps("4x0e+3x1e+4x1e+2x2e+3x2e")
    instr_uvu = [(0, 0, 0, 'uvu', True), (0, 1, 2, 'uvu', True), (1, 0, 1, 'uvu', True), (1, 1, 4, 'uvu', True), (2, 0, 3, 'uvu', True), (2, 1, 4, 'uvu', True)]
    # (two instructions write the same output slot 4 only if the multiplicities agree: 3x2e <- 1e x 1e (mul 3) ... keep the slots distinct instead)
    instr_uvu = [(0, 0, 0, 'uvu', True), (0, 1, 2, 'uvu', True), (1, 0, 1, 'uvu', True), (1, 1, 4, 'uvu', True), (2, 0, 3, 'uvu', True)]
    tp = o3.TensorProduct(ir1, ir2, iro, instr_uvu, internal_weights=False, shared_weights=False, path_normalization='none')
    x1, x2 = torch.randn(7, ir1.dim, generator=g), torch.randn(7, ir2.dim, generator=g)
    w = torch.randn(7, tp.weight_numel, generator=g)
    out["tp_uvu_instr"] = np.array([i[:3] for i in instr_uvu])
    out["tp_uvu_x1"], out["tp_uvu_x2"], out["tp_uvu_w"], out["tp_uvu_out"] = x1.numpy(), x2.numpy(), w.numpy(), tp(x1, x2, w).numpy()
    iro2 = o3.Irreps("5x0e+2x1e+3x2e")
    instr_uvw = [(i1, i2, io, 'uvw', True) for i1, (_, a) in enumerate(ir1) for i2, (_, b) in enumerate(ir2) for io, (_, c) in enumerate(iro2) if c in a * b]
    tp2 = o3.TensorProduct(ir1, ir2, iro2, instr_uvw, internal_weights=False, shared_weights=True, path_normalization='none')
    w2 = torch.randn(tp2.weight_numel, generator=g)
    out["tp_uvw_instr"] = np.array([i[:3] for i in instr_uvw])
    out["tp_uvw_w"], out["tp_uvw_out"] = w2.numpy(), tp2(x1, x2, w2).numpy()
    sections.append("tp")
except Exception as e:          # noqa: BLE001
    print("TensorProduct section skipped:", e)

# ---- the reference's own modules -------------------------------------------------------------------------------------------------------------
sys.path.insert(0, REF)


def _dump_state(prefix, module):
    for k, v in module.state_dict().items():
        out[f"{prefix}sd:{k}"] = v.detach().numpy()


try:
    from diffusion_edf.equiformer.graph_attention_transformer import SeparableFCTP
    torch.manual_seed(0)
    irr, sh = o3.Irreps("64x0e+32x1e+16x2e"), o3.Irreps("1x0e+1x1e+1x2e")
    m = SeparableFCTP(irr, sh, irr, fc_neurons=[128, 128, 64], use_activation=True, norm_layer=None, internal_weights=False).double().eval()
    x = torch.randn(11, irr.dim, generator=g)
    y = o3.spherical_harmonics(sh, torch.randn(11, 3, generator=g), normalize=True, normalization='component')
    s = torch.randn(11, 128, generator=g)
    with torch.no_grad():
        o = m(x, y, s)
    _dump_state("sfctp_", m)
    out["sfctp_x"], out["sfctp_y"], out["sfctp_s"], out["sfctp_out"] = x.numpy(), y.numpy(), s.numpy(), o.numpy()
    sections.append("sfctp")
except Exception as e:          # noqa: BLE001
    print("SeparableFCTP section skipped:", e)

try:
    from diffusion_edf.gnn_block import EquiformerBlock
    torch.manual_seed(1)
    irr, sh = o3.Irreps("64x0e+32x1e+16x2e"), o3.Irreps("1x0e+1x1e+1x2e")
    # the score head's block (multiscale_tensor_field.py:149-167: use_dst_feature=False -> no destination message, skip_1 = None)
    blk = EquiformerBlock(irreps_src=irr, irreps_dst=irr, irreps_edge_attr=sh, num_heads=4, fc_neurons=[128, 128, 64], irreps_mlp_mid=3, attn_type='mlp',
                          alpha_drop=0.0, proj_drop=0.0, drop_path_rate=0.0, use_dst_feature=False, skip_connection=True, bias=True,
                          use_src_point_attn=False, use_dst_point_attn=False, use_edge_weights=True).double().eval()
    n_src, n_dst, E = 9, 4, 30
    fs = torch.randn(n_src, irr.dim, generator=g)
    fd = torch.randn(n_dst, irr.dim, generator=g)
    es = torch.randint(0, n_src, (E,), generator=g)
    ed = torch.sort(torch.randint(0, n_dst - 1, (E,), generator=g)).values          # the last destination has no edge (empty segment)
    ea = o3.spherical_harmonics(sh, torch.randn(E, 3, generator=g), normalize=True, normalization='component')
    sc = torch.randn(E, 128, generator=g)
    lg = torch.randn(E, generator=g)
    from diffusion_edf.gnn_data import FeaturedPoints, GraphEdge
    z = lambda n: torch.zeros(n, dtype=torch.long)
    src = FeaturedPoints(x=torch.zeros(n_src, 3), f=fs, b=z(n_src))
    dst = FeaturedPoints(x=torch.zeros(n_dst, 3), f=fd, b=z(n_dst))
    edge = GraphEdge(edge_src=es, edge_dst=ed, edge_length=None, edge_attr=ea, edge_scalars=sc, edge_weights=None, edge_logits=lg)
    with torch.no_grad():
        o = blk(src_points=src, dst_points=dst, graph_edge=edge)
    _dump_state("block_", blk)
    out["block_fs"], out["block_fd"], out["block_es"], out["block_ed"] = fs.numpy(), fd.numpy(), es.numpy(), ed.numpy()
    out["block_ea"], out["block_sc"], out["block_lg"], out["block_out"] = ea.numpy(), sc.numpy(), lg.numpy(), o.f.numpy()
    sections.append("block")
except Exception as e:          # noqa: BLE001
    print("EquiformerBlock section skipped (needs torch_scatter / torch_cluster / edf_interface):", e)

out["sections"] = np.array(sections)
np.savez(os.path.join(HERE, "e3nn_0_4_4.npz"), **out)
print("wrote", os.path.join(HERE, "e3nn_0_4_4.npz"), "sections:", sections)
