"""Randomised parity sweep of the HIP path against the fp64 oracle (run on the GPU box: `python tests/stress_parity.py [n_cases] [seed]`).
Every case draws the model shape (lmax, number of scales and radii, radial-MLP width, time-MLP width, point attention), the cloud
sizes and the poses at random, so that tile raggedness, segment patterns (runs of equal destinations crossing the 16-lane rows and
the tile boundaries), empty neighbourhoods and the neighbour cap are hit in combinations the fixed tests do not enumerate."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import stage_check as SC
from diffusion_edf_amd import params, synthetic
from diffusion_edf_amd.gnn_data import FeaturedPoints


# DEDF_STRESS_LMAX3=1: lmax is drawn from {1, 2, 3} (the lmax-3 kernels exist for the default score-head / critic shapes in full precision);
# without it the sweep draws the cases of rounds 1-2 (lmax 1, 2) unchanged
LMAX_HI = 4 if os.environ.get("DEDF_STRESS_LMAX3") else 3


# DEDF_STRESS_QT=1: every third case carries query_time_encoding beside the edge time encoding, every third one query_time_encoding ALONE (the
# reference constructor's default); decided by the case index, so that the random stream -- and with it every earlier seed's cases -- is unchanged
STRESS_QT = bool(os.environ.get("DEDF_STRESS_QT"))


def draw_case(rng: np.random.Generator, max_lmax: int = 3, index: int = 0):
    lmax = min(int(rng.integers(1, LMAX_HI)), max_lmax)
    ns = int(rng.integers(1, 6))
    radii = sorted(float(r) for r in rng.uniform(2.0, 14.0, size=ns))
    inf_last = bool(rng.integers(0, 2))
    if inf_last:
        radii[-1] = None
    qt_mode = index % 3 if STRESS_QT else 0          # 1: query + edge time encoding, 2: query time encoding alone
    kw = synthetic.score_head_kwargs(lmax, radii=tuple(radii), query_time_encoding=qt_mode != 0, edge_time_encoding=qt_mode != 2)
    tf = kw['key_tensor_field_kwargs']
    tf['length_enc_max_r'] = 100. if inf_last else None
    shape = int(rng.integers(0, 4))
    if lmax == 3 or (qt_mode and lmax != 2) or (qt_mode == 2 and shape == 2):          # (the shapes instantiated with query_time_encoding)
        shape = 0
    if shape == 1 and lmax == 2:
        kw['time_emb_mlp'] = [512, 256, 128]
    elif shape == 2:
        tf['fc_neurons'] = [-1, 32, 32]
    point_attn = bool(rng.integers(0, 4) == 0)
    tf['use_src_point_attn'] = point_attn
    tf['r_mincut_nonscalar_sh'] = float(rng.uniform(0.05, 0.6))
    kw['max_time'] = float(rng.choice([1.0, 0.1]))
    cfg = params.HeadConfig.from_kwargs(kw)
    cfg.max_neighbors = int(rng.choice([1000, 1000, 1000, 3, 17]))
    seed = int(rng.integers(0, 1 << 30))
    P = params.init_params(cfg, seed=seed % 1000, randomize_all=True)
    n_scene = int(rng.integers(40, 2500))
    keys = synthetic.make_key_clouds(cfg, n_scene, seed=seed % 97)
    if point_attn:
        g = torch.Generator().manual_seed(seed)
        keys = [k._replace(w=torch.sigmoid(torch.randn(len(k.x), generator=g))) for k in keys]
    nQ = int(rng.integers(1, 130))
    query = synthetic.make_query(cfg, max(10 * nQ, 10), seed=seed % 89)
    query = FeaturedPoints(query.x[:nQ], query.f[:nQ], query.b[:nQ], query.w[:nQ])
    nT = int(rng.integers(1, 48))
    Ts = synthetic.make_poses(nT, seed=seed % 83, near_object=bool(rng.integers(0, 4) != 0))
    # keep the oracle's per-edge fp64 tensors (1568 wide) below ~1 GB: drop poses until the edge count fits
    import oracle.restatement as R
    def n_edges(T):
        xq = R.transform_points(query.x.double(), T).reshape(-1, 3)
        e = 0
        for k, r in zip(keys, cfg.radii):
            e += len(xq) * len(k.x) if r is None else int((torch.cdist(xq, k.x.double()) < r).sum())
        return e
    while len(Ts) > 1 and n_edges(Ts) > 60000:
        Ts = Ts[: max(1, len(Ts) // 2)]
    nT = len(Ts)
    time = torch.rand(nT, dtype=torch.float64, generator=torch.Generator().manual_seed(seed)) * 0.95 * kw['max_time'] + 0.02 * kw['max_time']
    time = time.float().double()          # (float32-representable: the time as the C ABI receives it, tests/stage_check.py::build_case)
    return kw, cfg, P, keys, query, Ts, time


AT_FP32_FLOOR = []          # DEDF_STRESS_QT: (case, kernel error, fp32 restatement error) of the query-time cases above 1e-4 whose fp32 restatement is too


def run_case(i, rng):
    kw, cfg, P, keys, query, Ts, time = draw_case(rng, index=i)
    max_nb = cfg.max_neighbors
    kw_o = dict(kw)
    ocfg_patch = {'max_neighbors': max_nb}
    # oracle
    import oracle.restatement as R
    ocfg = R.config_from_kwargs(kw)._replace(**ocfg_patch)
    Pd = R.cast_params(P, torch.float64)
    ok = [R.FeaturedPoints(k.x.double(), k.f.double(), k.b, None if k.w is None else k.w.double()) for k in keys]
    oq = R.FeaturedPoints(query.x.double(), query.f.double(), query.b, query.w.double())
    dbg = R.Debug()
    ang64, lin64 = R.score_head_forward(ocfg, Pd, Ts, ok, oq, time, dbg)
    # HIP path
    from diffusion_edf_amd.score_head import ScoreModelHead
    dev = torch.device('cuda:0')
    head = ScoreModelHead(**kw)
    head.cfg.max_neighbors = max_nb
    head.load_state_dict(P)
    head.to(dev)
    gk = [FeaturedPoints(k.x.to(dev), k.f.to(dev), k.b.to(dev), None if k.w is None else k.w.to(dev)) for k in keys]
    gq = FeaturedPoints(query.x.to(dev), query.f.to(dev), query.b.to(dev), query.w.to(dev))
    ang, lin = head(Ts.to(dev, torch.float32), gk, gq, time.to(dev, torch.float32))
    torch.cuda.synchronize()
    scale = float(max(ang64.abs().max(), lin64.abs().max()))
    err = max(float((ang.cpu().double() - ang64).abs().max()), float((lin.cpu().double() - lin64).abs().max())) / scale
    edges_ok = head.stats()['n_edges'] == dbg['n_edges_per_scale']
    if not edges_ok:
        # a pair within 1e-7 of a radius can fall on either side in fp32 and fp64 (its weight is ~0 there): the fp32 restatement decides
        d32 = R.Debug()
        ok32 = [R.FeaturedPoints(k.x, k.f, k.b, k.w) for k in keys]
        R.score_head_forward(ocfg, R.cast_params(P, torch.float32), Ts.float(), ok32, R.FeaturedPoints(query.x, query.f, query.b, query.w), time.float(), d32)
        edges_ok = head.stats()['n_edges'] == d32['n_edges_per_scale']
        print(f"   (edge count differs from the fp64 restatement {dbg['n_edges_per_scale']}; fp32 restatement {d32['n_edges_per_scale']})")
    desc = (f"lmax {cfg.lmax} radii {cfg.radii} fc {cfg.fc_neurons} temb {cfg.time_emb_mlp[0]} pattn {cfg.use_src_point_attn} cap {max_nb} "
            f"keys {[len(k.x) for k in keys]} nQ {len(query.x)} nT {len(Ts)} E {dbg['n_edges_per_scale']}")
    if err >= 1e-4 and edges_ok:
        # ill-conditioned case?  The reference computes in fp32: if ITS arithmetic (the fp32 restatement) is as far from fp64 as the
        # kernel is, and the kernel agrees with the fp32 restatement to 2e-5, the case says nothing about the kernel (seen once in 200
        # cases: seed 101 case 182, fp32 restatement 2.6e-4 from fp64, kernel 6.7e-6 from the fp32 restatement)
        ok32 = [R.FeaturedPoints(k.x, k.f, k.b, k.w) for k in keys]
        a32, l32 = R.score_head_forward(ocfg, R.cast_params(P, torch.float32), Ts.float(), ok32, R.FeaturedPoints(query.x, query.f, query.b, query.w), time.float())
        gap = max(float((a32.double() - ang64).abs().max()), float((l32.double() - lin64).abs().max())) / scale
        e32 = max(float((ang.cpu().double() - a32.double()).abs().max()), float((lin.cpu().double() - l32.double()).abs().max())) / scale
        print(f"   (fp32 restatement vs fp64: {gap:.2e}; kernel vs fp32 restatement: {e32:.2e})")
        if e32 < 2e-5 and gap > 0.5 * err:
            err = e32
        elif kw.get('query_time_encoding') and err < 1.25 * gap:
            # query_time_encoding with randomised weights: the reference's OWN fp32 arithmetic is outside 1e-4 of fp64 here (the time-dependent destination
            # message, tests/test_gpu_parity.py::test_query_time_encoding) and the kernel is no farther than it.  Counted separately, never as "within 1e-4".
            AT_FP32_FLOOR.append((i, err, gap))
            print(f"   (query-time case at the fp32 restatement's own floor: kernel {err:.2e}, restatement {gap:.2e} from fp64)")
    if err >= 1e-4 and edges_ok and any(r is not None and len(k.x) > max_nb for k, r in zip(keys, cfg.radii)):
        # a key within fp32 rounding of a radius UNDER A BINDING NEIGHBOUR CAP?  Whether that pair is a neighbour is decided by the last bit of an fp32
        # distance (the reference's torch_cluster kernel has the same ambiguity); without a cap it would only add or drop an edge of weight ~0, with
        # the cap it decides whether ANOTHER, fully weighted key takes the 17th place -- same edge count, other edge set (seen once in 120 cases:
        # seed 505 case 23, a key at r (1 - 1.7e-7); tests/probe/stress_edges.py lists the pairs).  The oracle with the radii moved by 3e-7 either
        # way decides: the kernel must agree with one of them.
        for f in (1.0 - 3e-7, 1.0 + 3e-7):
            oc2 = ocfg._replace(r_cluster_multiscale=[None if r is None else r * f for r in ocfg.r_cluster_multiscale])
            a2, l2 = R.score_head_forward(oc2, Pd, Ts, ok, oq, time)
            s2 = float(max(a2.abs().max(), l2.abs().max()))
            e2 = max(float((ang.cpu().double() - a2).abs().max()), float((lin.cpu().double() - l2).abs().max())) / s2
            print(f"   (binding neighbour cap; oracle with the radii x {f}: kernel {e2:.2e} from it)")
            if e2 < 1e-4:
                err = e2
                break
    print(f"case {i:3d} err {err:.2e} edges_ok {edges_ok}  {desc}", flush=True)
    return err, edges_ok, desc


def run_sample_case(i, rng):
    """ScoreModelBase.sample (3-5 Langevin steps, injected noise) against the oracle's float64 loop with an fp32 score"""
    kw, cfg, P, keys, query, Ts, time = draw_case(rng, index=i)
    import oracle.restatement as R
    from diffusion_edf_amd.score_head import ScoreModelHead
    from diffusion_edf_amd.score_model_base import ScoreModelBase
    Ts = Ts[:12]
    n_steps = [int(rng.integers(1, 4)), int(rng.integers(1, 3))]
    tmax = kw['max_time']
    sched = [[tmax, 0.4 * tmax], [0.4 * tmax, 0.1 * tmax]]
    dts = [float(rng.uniform(0.005, 0.05)), float(rng.uniform(0.005, 0.03))]
    temps = [float(rng.uniform(0.0, 1.5)), float(rng.uniform(0.0, 1.0))]
    g = torch.Generator().manual_seed(int(rng.integers(0, 1 << 30)))
    noise = torch.randn(sum(n_steps), 2, len(Ts), 3, generator=g, dtype=torch.float64)
    ocfg = R.config_from_kwargs(kw)._replace(max_neighbors=cfg.max_neighbors)
    ok = [R.FeaturedPoints(k.x, k.f, k.b, k.w) for k in keys]
    oq = R.FeaturedPoints(query.x, query.f, query.b, query.w)
    ref = R.sample(ocfg, P, Ts, ok, oq, sched, n_steps, dts, temperatures=temps, noise=noise)
    dev = torch.device('cuda:0')
    head = ScoreModelHead(**kw)
    head.cfg.max_neighbors = cfg.max_neighbors
    head.load_state_dict(P)
    head.to(dev)
    gk = [FeaturedPoints(k.x.to(dev), k.f.to(dev), k.b.to(dev), None if k.w is None else k.w.to(dev)) for k in keys]
    gq = FeaturedPoints(query.x.to(dev), query.f.to(dev), query.b.to(dev), query.w.to(dev))
    out = ScoreModelBase(head).sample(Ts.to(dev), gk, gq, sched, n_steps, dts, temperatures=temps, noise=noise).cpu()
    err = float((out - ref).abs().max())
    move = float((ref[-1] - ref[0]).abs().max())
    # ASSERTED (round 6): (a) the score itself at the SEED poses (first step's time), kernel against the fp64 restatement, within north_star's 1e-4 of the
    # score scale; (b) the poses after the FIRST Langevin step within 1e-4.  LOGGED only: the pose difference after all 2-5 steps -- a sampler amplifies a
    # rounding-level difference by 3-10x per step where large steps make the trajectory chaotic (profiles/r05zy_stress_sample.log: two of 40 cases end at
    # 1.7e-3 / 3.4e-3 from a seed-score error of 8.1e-5 / 2.7e-5 where the fp32 restatement itself is 5.6e-5 / 2.0e-5 away).
    t0 = torch.full((len(Ts),), sched[0][0], dtype=torch.float64)
    k64 = [R.FeaturedPoints(k.x.double(), k.f.double(), k.b, None if k.w is None else k.w.double()) for k in keys]
    q64 = R.FeaturedPoints(query.x.double(), query.f.double(), query.b, query.w.double())
    a64, l64 = R.score_head_forward(ocfg, R.cast_params(P, torch.float64), Ts, k64, q64, t0)
    ag, lg = head(Ts.to(dev).float(), gk, gq, t0.to(dev).float())
    sc = float(max(a64.abs().max(), l64.abs().max()))
    seed_err = max(float((ag.cpu().double() - a64).abs().max()), float((lg.cpu().double() - l64).abs().max())) / sc
    step1 = float((out[1] - ref[1]).abs().max())
    seed_bar = 1e-4
    if seed_err >= 1e-4 and kw.get('query_time_encoding'):      # (DEDF_STRESS_QT) the reference's own fp32 arithmetic as the floor, as in run_case -- and said so
        a32, l32 = R.score_head_forward(ocfg, R.cast_params(P, torch.float32), Ts.float(), ok, oq, t0.float())
        gap = max(float((a32.double() - a64).abs().max()), float((l32.double() - l64).abs().max())) / sc
        print(f"   (query-time case: fp32 restatement {gap:.2e} from fp64 at the seed poses, kernel {seed_err:.2e})")
        if seed_err < 1.25 * gap:
            AT_FP32_FLOOR.append((i, seed_err, gap)); seed_bar = 1.25 * gap
    ok_ = out.shape == ref.shape and bool(torch.isfinite(out).all()) and seed_err < seed_bar and step1 < 1e-4
    print(f"sample {i:3d} score at the seed poses {seed_err:.2e}, after step 1 {step1:.2e}; after all steps |dT| {err:.2e} (poses moved {move:.2e}) steps {n_steps} nT {len(Ts)} "
          f"lmax {cfg.lmax} radii {cfg.radii} cap {cfg.max_neighbors}", flush=True)
    if err > 2e-4:      # where along the trajectory the difference appears: a step that starts it, or growth from the rounding level
        print("   per step:", [f"{float((out[k] - ref[k]).abs().max()):.1e}" for k in range(len(out))],
              "worst pose per step:", [int((out[k] - ref[k]).abs().amax(dim=-1).argmax()) for k in range(len(out))], flush=True)
        a32, l32 = R.score_head_forward(ocfg, R.cast_params(P, torch.float32), Ts.float(), ok, oq, t0.float())
        print(f"   fp32 restatement at the seed poses: {max(float((a32.double() - a64).abs().max()), float((l32.double() - l64).abs().max())) / sc:.2e} from the fp64 restatement; "
              f"per pose (kernel): {[f'{float(max((ag.cpu().double() - a64)[p].abs().max(), (lg.cpu().double() - l64)[p].abs().max())) / sc:.1e}' for p in range(len(Ts))]}", flush=True)
    return err, ok_, ""


def run_ebm_case(i, rng):
    """EbmScoreModelHead.compute_energy (the critic) on random shapes; every second case in half-precision GEMM mode"""
    import oracle.restatement as R
    from diffusion_edf_amd.score_head import EbmScoreModelHead
    lmax = int(rng.integers(1, LMAX_HI))
    ns = int(rng.integers(1, 5))
    radii = tuple(sorted(float(r) for r in rng.uniform(2.5, 10.0, size=ns)))
    kw = synthetic.ebm_head_kwargs(lmax, radii=radii)
    kw['key_tensor_field_kwargs']['r_mincut_nonscalar_sh'] = float(rng.uniform(0.05, 0.6))
    cfg = params.HeadConfig.from_kwargs(kw)
    seed = int(rng.integers(0, 1 << 30))
    P = params.init_params(cfg, seed=seed % 1000, randomize_all=True)
    keys = synthetic.make_key_clouds(cfg, int(rng.integers(60, 1500)), seed=seed % 97)
    nQ = int(rng.integers(1, 100))
    query = synthetic.make_query(cfg, max(10 * nQ, 10), seed=seed % 89)
    query = FeaturedPoints(query.x[:nQ], query.f[:nQ], query.b[:nQ], query.w[:nQ])
    Ts = synthetic.make_poses(int(rng.integers(1, 40)), seed=seed % 83, near_object=True)
    time = torch.ones(len(Ts), dtype=torch.float64)
    ok = [R.FeaturedPoints(k.x.double(), k.f.double(), k.b) for k in keys]
    oq = R.FeaturedPoints(query.x.double(), query.f.double(), query.b, query.w.double())
    e64 = R.compute_energy(R.config_from_kwargs(kw), R.cast_params(P, torch.float64), Ts, ok, oq, time)
    dev = torch.device('cuda:0')
    half = bool(i % 2) and lmax < 3
    head = EbmScoreModelHead(**{k: v for k, v in kw.items() if k != 'ebm'})
    head.load_state_dict(P)
    head.to(dev)
    if half:
        head.half()
    gk = [FeaturedPoints(k.x.to(dev), k.f.to(dev), k.b.to(dev)) for k in keys]
    gq = FeaturedPoints(query.x.to(dev), query.f.to(dev), query.b.to(dev), query.w.to(dev))
    e = head.compute_energy(Ts.to(dev).float(), gk, gq, time.to(dev).float()).cpu().double()
    err = float((e - e64).abs().max() / e64.abs().max())
    ok_ = err < (5e-3 if half else 1e-4)
    print(f"ebm {i:3d} err {err:.2e} half {half} lmax {lmax} radii {[round(r, 2) for r in radii]} keys {[len(k.x) for k in keys]} nQ {nQ} nT {len(Ts)}", flush=True)
    return err, ok_, ""


def run_half_case(i, rng):
    """score head in half-precision GEMM mode (model.half()): stated tolerance 5e-3 of the score scale"""
    kw, cfg, P, keys, query, Ts, time = draw_case(rng, max_lmax=2)
    import oracle.restatement as R
    from diffusion_edf_amd.score_head import ScoreModelHead
    ocfg = R.config_from_kwargs(kw)._replace(max_neighbors=cfg.max_neighbors)
    ok = [R.FeaturedPoints(k.x.double(), k.f.double(), k.b, None if k.w is None else k.w.double()) for k in keys]
    oq = R.FeaturedPoints(query.x.double(), query.f.double(), query.b, query.w.double())
    ang64, lin64 = R.score_head_forward(ocfg, R.cast_params(P, torch.float64), Ts, ok, oq, time)
    dev = torch.device('cuda:0')
    head = ScoreModelHead(**kw)
    head.cfg.max_neighbors = cfg.max_neighbors
    head.load_state_dict(P)
    head.to(dev)
    head.half()
    gk = [FeaturedPoints(k.x.to(dev), k.f.to(dev), k.b.to(dev), None if k.w is None else k.w.to(dev)) for k in keys]
    gq = FeaturedPoints(query.x.to(dev), query.f.to(dev), query.b.to(dev), query.w.to(dev))
    ang, lin = head(Ts.to(dev, torch.float32), gk, gq, time.to(dev, torch.float32))
    scale = float(max(ang64.abs().max(), lin64.abs().max()))
    err = max(float((ang.cpu().double() - ang64).abs().max()), float((lin.cpu().double() - lin64).abs().max())) / scale
    print(f"half {i:3d} err {err:.2e} lmax {cfg.lmax} fc {cfg.fc_neurons} radii {cfg.radii} nQ {len(query.x)} nT {len(Ts)}", flush=True)
    return err, err < 5e-3, ""


def run_cases(n, seed):
    rng = np.random.default_rng(seed)
    bad = []
    for i in range(n):
        err, eok, desc = run_case(i, rng)
        if not (err < 1e-4 and eok):
            bad.append((i, err, eok, desc))
    return bad


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    if len(sys.argv) > 3:
        fn = {"sample": run_sample_case, "ebm": run_ebm_case, "half": run_half_case}[sys.argv[3]]
        rng = np.random.default_rng(seed)
        res = [fn(i, rng) for i in range(n)]
        nbad = sum(1 for r in res if not r[1])
        print("FAILED" if nbad else "ALL OK", nbad, "of", n, "largest difference", max(r[0] for r in res),
              f"({len(AT_FP32_FLOOR)} query-time cases at the fp32 restatement's own floor: {[(i, float(f'{e:.2e}'), float(f'{g:.2e}')) for i, e, g in AT_FP32_FLOOR]})" if AT_FP32_FLOOR else "")
        sys.exit(1 if nbad else 0)
    rng = np.random.default_rng(seed)
    bad = []
    for i in range(n):
        err, eok, desc = run_case(i, rng)
        if not (err < 1e-4 and eok) and not any(f[0] == i for f in AT_FP32_FLOOR):
            bad.append((i, err, eok, desc))
    print("FAILED" if bad else "ALL OK", len(bad), "of", n, f"({n - len(bad) - len(AT_FP32_FLOOR)} within 1e-4, {len(AT_FP32_FLOOR)} query-time cases at the fp32 restatement's own floor: "
          f"{[(i, float(f'{e:.2e}'), float(f'{g:.2e}')) for i, e, g in AT_FP32_FLOOR]})" if AT_FP32_FLOOR else "")
    for b in bad:
        print(b)
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
