"""The e3nn convention pin as a switch: compares the oracle -- and the constants the product's kernels are generated from -- with
outputs of e3nn 0.4.4 and of the reference's own modules stored in tests/golden/e3nn_0_4_4.npz (written by tests/golden/make_e3nn_golden.py in an
environment that has e3nn).  SKIPPED while that file is absent (e3nn is not importable in the build container: no wheel, no network): the day the
file is committed these tests run and the oracle's e3nn restatement is pinned by reference outputs, with no other change.

What each section would pin:  w3j -- the per-block sign and normalisation of the real 3j symbols;  sh -- sign / order / normalisation of the real
spherical harmonics;  n2m -- the normalize2mom constants;  tp -- the sqrt(2 l_out + 1) path factor and the flat weight layout of o3.TensorProduct;
sfctp / block -- a whole SeparableFCTP / EquiformerBlock of the reference with its own state dict."""
import os

import numpy as np
import pytest
import torch

from diffusion_edf_amd import so3 as pso3
from oracle import restatement as R
from oracle import so3_oracle as oso3

FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "e3nn_0_4_4.npz")
pytestmark = pytest.mark.skipif(not os.path.exists(FIX), reason="tests/golden/e3nn_0_4_4.npz absent: run tests/golden/make_e3nn_golden.py where e3nn==0.4.4 is importable "
                                                                "(parity of the e3nn restatement stays UNPINNED until then)")


@pytest.fixture(scope="module")
def gold():
    return np.load(FIX, allow_pickle=False)


def _need(gold, section):
    if section not in set(gold["sections"].tolist()):
        pytest.skip(f"the fixture holds no `{section}` section")


def test_generator_script_is_committed_next_to_the_fixture():
    assert os.path.exists(os.path.join(os.path.dirname(FIX), "make_e3nn_golden.py"))


def test_wigner_3j_blocks_sign_and_normalisation(gold):
    _need(gold, "w3j")
    for l1 in range(4):
        for l2 in range(4):
            for l3 in range(abs(l1 - l2), min(3, l1 + l2) + 1):
                ref = gold[f"w3j_{l1}{l2}{l3}"]
                assert np.allclose(oso3.w3j(l1, l2, l3), ref, atol=1e-12), (l1, l2, l3, "oracle/so3_oracle.py")
                assert np.allclose(pso3.wigner_3j(l1, l2, l3), ref, atol=1e-12), (l1, l2, l3, "diffusion_edf_amd/so3.py (source of csrc/dedf_tables.h)")


def test_spherical_harmonics_sign_order_normalisation(gold):
    _need(gold, "sh")
    x = gold["sh_points"]
    u = x / np.linalg.norm(x, axis=-1, keepdims=True)
    for l in range(4):
        assert np.allclose(oso3.sh(l, u), gold[f"sh_{l}"], atol=1e-12), (l, "oracle")
        assert np.allclose(pso3.spherical_harmonics(l, x), gold[f"sh_{l}"], atol=1e-12), (l, "product tables")
    got = R.spherical_harmonics([(1, 0), (1, 1), (1, 2), (1, 3)], torch.tensor(x))
    assert np.allclose(got.numpy(), np.concatenate([gold[f"sh_{l}"] for l in range(4)], -1), atol=1e-12)


def test_normalize2mom_constants(gold):
    _need(gold, "n2m")
    assert abs(oso3.C_SILU - float(gold["n2m_silu"])) < 1e-12 and abs(oso3.C_SIGMOID - float(gold["n2m_sigmoid"])) < 1e-12
    assert abs(oso3.C_SLRELU - float(gold["n2m_slrelu"])) < 1e-12


def test_tensor_product_path_factor_and_weight_layout(gold):
    _need(gold, "tp")
    ir1, ir2 = [(4, 0), (3, 1), (2, 2)], [(1, 0), (1, 1), (1, 2)]
    iro = [(4, 0), (3, 1), (4, 1), (2, 2), (3, 2)]
    x1, x2 = torch.tensor(gold["tp_uvu_x1"]), torch.tensor(gold["tp_uvu_x2"])
    tp = R.TensorProduct(ir1, ir2, iro, [R.TPInstr(int(a), int(b), int(c), 'uvu') for a, b, c in gold["tp_uvu_instr"]])
    assert np.allclose(tp(x1, x2, torch.tensor(gold["tp_uvu_w"])).numpy(), gold["tp_uvu_out"], atol=1e-10)
    iro2 = [(5, 0), (2, 1), (3, 2)]
    tp2 = R.TensorProduct(ir1, ir2, iro2, [R.TPInstr(int(a), int(b), int(c), 'uvw') for a, b, c in gold["tp_uvw_instr"]])
    assert np.allclose(tp2(x1, x2, torch.tensor(gold["tp_uvw_w"])).numpy(), gold["tp_uvw_out"], atol=1e-10)


def _state(gold, prefix):
    pre = prefix + "sd:"
    return {k[len(pre):]: torch.tensor(gold[k]) for k in gold.files if k.startswith(pre)}


def test_separable_fctp_of_the_reference(gold):
    """graph_attention_transformer.py:60-135: RadialProfile -> depth-wise TP -> LinearRS -> Gate, with the module's own state dict"""
    _need(gold, "sfctp")
    irr, sh = R.parse_irreps("64x0e+32x1e+16x2e"), R.parse_irreps("1x0e+1x1e+1x2e")
    P = _state(gold, "sfctp_")
    x, y, s_ = (torch.tensor(gold[k]) for k in ("sfctp_x", "sfctp_y", "sfctp_s"))
    dtp, dtp_simpl, lin_out, gate = R.separable_fctp_dtp_lin(irr, sh, irr, True)
    w = R.radial_profile(s_, P, "dtp_rad", 3)
    out = R.gate(R.linear_rs(dtp(x, y, w), dtp_simpl, lin_out, P, "lin"), *gate)
    ref = torch.tensor(gold["sfctp_out"])
    assert float((out - ref).abs().max()) < 1e-9 * max(1.0, float(ref.abs().max()))


def test_equiformer_block_of_the_reference(gold):
    """gnn_block.py:164-218 with use_dst_feature=False (the score head's block), incl. a destination without edges (empty softmax segment)"""
    _need(gold, "block")
    cfg = R.config_from_kwargs(__import__("diffusion_edf_amd.synthetic", fromlist=["x"]).score_head_kwargs(2))
    P = {"blk." + k: v for k, v in _state(gold, "block_").items()}
    t = lambda k: torch.tensor(gold[k])
    out, _ = R.equiformer_block(cfg, P, "blk", t("block_fs"), t("block_es"), t("block_ed"), t("block_ea"), t("block_sc"), t("block_lg"), len(gold["block_fd"]))
    ref = t("block_out")
    assert float((out - ref).abs().max()) < 1e-9 * max(1.0, float(ref.abs().max()))
