#!/usr/bin/env python
"""bench.py — the headline benchmark of BASELINE.json on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...)

Workload (config C2 of BASELINE.md, the one the metric is quoted on): synthetic 4096-pt scene -> key clouds 820/164/33/7,
1024-pt grasp -> 103 query points, lmax 2 (64x0e+32x1e+16x2e, full irreps TP + graph attention), 1000 poses PER GPU
(weak scaling: config C3 = 8 x 1000 poses), random-init weights of the reference architecture, inputs resident in HBM.
A "step" is one denoising step of the fused sampler over the rank's pose batch: neighbour search + score forward (f32) +
Langevin update (f64).  value = (poses on all ranks) x K / max-over-ranks wall time  [pose-steps/s].
One RCCL all-gather of the final poses closes the timed region.

Extra objects on the JSON line:
  roofline      dominant kernel (fused per-edge pipeline k_edge).  ONE definition: achieved = 2 x 193 344 MAC (SURVEY section 8(d)'s algorithmic
                constant at lmax 2, dense-CG convention; 105 536 / 338 304 at lmax 1 / 3) x edges of the launch / the kernel's HIP-event duration;
                peak = 2500 / 3 TFLOP/s -- every dense GEMM is a 3-term split-fp16 product (hi*hi + hi*lo + lo*hi, fp32 accumulate) on the
                2.5 PFLOP/s dense fp16 MFMAs of MI355X_MICROARCH.md --; frac = achieved / peak.  "bound": "mfma" names that roof; what limits the
                kernel is the instruction issue of its one wave per SIMD ("limiter").  Beside it, the hardware view: `frac_mfma_issued`
                (SQ_INSTS_MFMA x 32 768 FLOP / launch time / 2.5 PFLOP/s from the newest PMC profile) and `frac_gemm_executed` (the same quantity from
                the kernel's own term count, edge_frame_gemm_mac(): the two agree to 3 %).  `frac_round2_4_units` is the edge rate in the unit the
                first rounds were judged in (152 384 MAC per edge), kept for comparison only.  traffic = HBM bytes per launch from the newest
                profiles/*pmc*.json (`traffic_source` names it).  roofline.radial_table: the sampler tabulates the front of the radial network per
                step (every pose shares the time); the object also carries the per-edge reading of the same K steps.
  cpu_baseline  the CPU restatement oracle ("port", fp32, same inputs) timed on this host on a bounded pose sample: 1 thread, 16
                threads and all physical cores (lscpu).
  config.score_fwd_ms_at_t0.5   one score evaluation (no Langevin update) of the seeded poses at the fixed time t = 0.5.
  config.feature_extractors_ms  the step before the path, outside the timed region (N = 1): UnetFeatureExtractor on the scene cloud,
                                KeypointExtractor on the grasp cloud (random-init weights of the shipped panda shapes); ms per forward.
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import re
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

M_EDGE = {1: 105_536, 2: 193_344, 3: 338_304}          # algorithmic MAC per edge (dense-CG convention; lmax 3: tests/test_lmax3.py re-derives it)
M_NODE = {1: 56_320 + 186_560, 2: 68_352 + 343_264, 3: 73_024 + 428_784}
M_EDGE_FRONT = {1: 40_960, 2: 40_960, 3: 40_960}      # of which: edge pre-linear 128x128 + RadialProfile layers 1 (128x128) and 2 (128x64): the part the sampler's
                                           # radial table evaluates per length-grid node (40 k nodes per step) instead of per edge (2 M)
M_EDGE_CG = {1: 2 * 2_080, 2: 2 * 14_880, 3: 2 * 55_200}   # of which: the two depth-wise TPs (dense-CG convention), VALU work
PEAK_FP32_MFMA_TFLOPS = 157.3             # dense fp32 MFMA = fp32 vector peak (MI355X_MICROARCH.md)
PEAK_FP16_MFMA_TFLOPS = 2500.0            # dense fp16 MFMA


def mix_peak_tflops(lmax):
    """bound of k_edge's instruction mix in fp32-equivalent FLOP/s: GEMMs as 3 fp16 MFMA products, CG on the fp32 VALU"""
    m, mc = M_EDGE[lmax], M_EDGE_CG[lmax]
    t = (m - mc) * 3.0 / PEAK_FP16_MFMA_TFLOPS + mc / PEAK_FP32_MFMA_TFLOPS
    return m / t


def build_inputs(lmax, n_scene, n_grasp, n_poses, first_pose, device):
    from diffusion_edf_amd import params, synthetic
    from diffusion_edf_amd.gnn_data import FeaturedPoints
    kw = synthetic.score_head_kwargs(lmax)
    cfg = params.HeadConfig.from_kwargs(kw)
    P = params.init_params(cfg, seed=2)
    keys = synthetic.make_key_clouds(cfg, n_scene, seed=0)
    query = synthetic.make_query(cfg, n_grasp, seed=0)
    Ts = synthetic.make_poses(n_poses, seed=1, first_pose_index=first_pose)
    if device is not None:
        keys = [FeaturedPoints(k.x.to(device), k.f.to(device), k.b.to(device)) for k in keys]
        query = FeaturedPoints(query.x.to(device), query.f.to(device), query.b.to(device), query.w.to(device))
        Ts = Ts.to(device)
    return kw, cfg, P, keys, query, Ts


def edge_frame_gemm_mac(lmax, table_on=True, f0=128, h1=128, h2=64):
    """fp32-equivalent multiply-adds per edge of the dense GEMMs the edge-aligned-frame kernels ISSUE, in the shapes they issue them in.  Every GEMM
    term is one MFMA triple (hi*hi + hi*lo + lo*hi; one MFMA in half-precision mode) of 32 output rows x 16 K-channels x 32 edges = 512 MAC per edge,
    padding of narrower blocks into the 32-row tile included (dedf_net.h: make_dtp_walk_so2 / make_sval_walk; 8x3e runs as 16x3e):
      last radial layer   ceil(weight_numel / 32) row tiles x H2 / 16 K-chunks
      scalar group        K-chunks of the l3 = 0 paths x r0_tiles (lin scalars + gates | sep_alpha rows)
      l3 >= 1 group       one triple per (path, K-chunk, reachable output component)
      value               the l3 = 0 paths into two tiles + the same (path, K-chunk, component) terms
      front (table off)   pre-linear + radial layers 1, 2
    lmax 2: (60 + 42 + 53 + 67) x 512 = 113 664; the PMC count of the timed kernel (SQ_INSTS_MFMA / 3 x 512 / edges) is 111 k (profiles/r05zi_pmc_summary.json)."""
    from diffusion_edf_amd import so2
    mul = lambda l: 16 if l >= 3 else 64 >> l
    paths = [(a, b, c) for a in range(lmax + 1) for b in range(lmax + 1) for c in range(abs(a - b), min(lmax, a + b) + 1)]
    wn = sum(mul(p[0]) for p in paths)
    lin0_rows = 64 + sum(mul(l) for l in range(1, lmax + 1))
    nr0 = ((lin0_rows + 31) // 32 * 32 + 64) // 32
    t_l3 = (wn + 31) // 32 * (h2 // 16)
    t_0 = sum(mul(p[0]) // 16 for p in paths if p[2] == 0) * nr0
    t_ge1 = sum((mul(p[0]) // 16) * len(so2.so2_terms(*p)) for p in paths if p[2] >= 1)
    t_val = sum((mul(p[0]) // 16) * 2 for p in paths if p[2] == 0) + t_ge1
    t_front = 0 if table_on else (f0 // 32) * 4 + (h1 // 32) * (f0 // 16) + (h2 // 32) * (h1 // 16)
    return 512 * (t_l3 + t_0 + t_ge1 + t_val + t_front)


def build_config5(n_scene, n_grasp, device, reps=3):
    """BASELINE config 5 as ONE model (synthetic.config5_model_kwargs: reference multiscale_score_model.py:27-112 at lmax 3): the scene cloud goes
    through the UNet key model, the grasp cloud through the KeypointExtractor query model -- both on the HIP kernels, seeded random-init weights --
    and the sampler runs on what they return.  Also returns the extractors' ms per forward (outside the timed region)."""
    import numpy as np
    from diffusion_edf_amd import agent, synthetic
    from diffusion_edf_amd.gnn_data import FeaturedPoints
    model = agent.MultiscaleScoreModel(**synthetic.config5_model_kwargs(), deterministic=True).to(device).eval()

    def cloud(x, seed):
        x = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(device)
        f = torch.rand(len(x), 3, generator=torch.Generator().manual_seed(seed)).to(device)
        return FeaturedPoints(x=x, f=f, b=torch.zeros(len(x), dtype=torch.long, device=device), w=None)

    scene, grasp = cloud(synthetic.make_scene(n_scene, seed=0), 0), cloud(synthetic.make_grasp(n_grasp, seed=0), 1)

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            out = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3, out

    ms_k, keys = timed(lambda: model.get_key_pcd_multiscale(scene))
    ms_q, query = timed(lambda: model.get_query_pcd(grasp))
    extract = {"unet_scene": ms_k, "unet_levels": [len(k.x) for k in keys], "keypoint_grasp": ms_q, "keypoints": len(query.x),
               "irreps": "64x0e+32x1e+16x2e+8x3e", "note": "the model's own key / query extractors at lmax 3 on the workload's clouds"}
    return model, keys, query, extract


def physical_cores():
    """physical cores of this host from lscpu (sockets x cores per socket); None when lscpu is missing"""
    try:
        import subprocess
        txt = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        g = lambda k: int(re.search(rf"^{k}:\s*(\d+)", txt, re.M).group(1))
        return g(r"Socket\(s\)") * g(r"Core\(s\) per socket")
    except Exception:
        return None


def cpu_baseline(lmax, n_scene, n_grasp, n_sample_poses=32, t=0.5, budget_s=8.0):
    """oracle (kind 'port'): one score evaluation + Langevin step on a bounded pose sample of the same inputs, fp32.
    Legs: 1 thread, 16 threads, and all physical cores of the host (lscpu); `value` is the best leg and `cores` the
    threads it used.  Every leg runs whole repetitions for ~budget_s seconds (at least 2)."""
    from oracle import restatement as R
    kw, cfg, P, keys, query, Ts = build_inputs(lmax, n_scene, n_grasp, n_sample_poses, 0, None)
    ocfg = R.config_from_kwargs(kw)
    ok = [R.FeaturedPoints(k.x, k.f, k.b) for k in keys]
    oq = R.FeaturedPoints(query.x, query.f, query.b, query.w)
    noise = torch.zeros(1, 2, n_sample_poses, 3, dtype=torch.float64)
    phys = physical_cores() or torch.get_num_threads()
    prev = torch.get_num_threads()
    legs = {}
    try:
        for n_thr in sorted({1, min(16, phys), phys}):
            torch.set_num_threads(n_thr)
            n_p = min(4, max(1, n_sample_poses // 4)) if n_thr == 1 else n_sample_poses          # the single-thread leg runs a smaller sample (same inputs)
            R.sample(ocfg, P, Ts[:2], ok, oq, [[t, t]], [1], [0.04], noise=noise[:, :, :2])          # warm-up (builds caches)
            reps, t0 = 0, time.perf_counter()
            while True:
                R.sample(ocfg, P, Ts[:n_p], ok, oq, [[t, t]], [1], [0.04], noise=noise[:, :, :n_p])
                reps += 1
                el = time.perf_counter() - t0
                if (el > budget_s and reps >= 2) or reps >= 50:
                    break
            legs[n_thr] = dict(value=n_p * reps / el, poses=n_p, reps=reps, seconds=round(el, 2))
    finally:
        torch.set_num_threads(prev)
    best = max(legs, key=lambda k: legs[k]["value"])
    return dict(value=legs[best]["value"], unit="pose-steps/s", cores=best, kind="port",
                physical_cores=phys, single_thread_value=legs[1]["value"],
                legs={str(k): v for k, v in legs.items()},
                sample=f"1 denoise step at t={t} of the same inputs ({n_scene}-pt scene, lmax {lmax}), fp32 CPU restatement (oracle/restatement.py): "
                       + "; ".join(f"{k} thread(s): {v['reps']} x {v['poses']} poses in {v['seconds']} s" for k, v in legs.items()))


def radial_table_note(args, flops, e_per_launch, edge_ms, peak, per_edge=None):
    """`achieved` / `frac` count the ALGORITHMIC FLOP of the reference per edge.  In the sampler every pose of a step shares the
    time, so the front of the radial network is a function of (scale, edge length): the default path evaluates it once per step on a length
    grid and interpolates per edge -- 40 960 of the 193 344 MAC/edge are then not executed per edge.  This object says so and gives the
    fraction on the FLOP the edge kernel really executes."""
    on = (args.lmax in (2, 3) and not args.half and not args.no_radial_table)
    if not on:
        return {"enabled": False}
    ex = 2.0 * e_per_launch * (M_EDGE[args.lmax] - M_EDGE_FRONT[args.lmax])
    ach = ex / (edge_ms * 1e-3) / 1e12 if edge_ms > 0 else 0.0
    pe = None
    if per_edge is not None:      # the per-edge reading of the same K steps (timed like `value`): frac on the same algorithmic FLOP, all of them executed
        pe = dict(per_edge)
        pe["frac"] = (flops / (per_edge["k_edge_ms"] * 1e-3) / 1e12) / peak if per_edge["k_edge_ms"] > 0 else 0.0
    return {"enabled": True, "without_table": pe, "executed_flop_per_launch": ex, "achieved_on_executed_flop": ach, "frac_on_executed_flop": ach / peak,
            "what": "front of the radial network (length encoding, edge pre-linear, RadialProfile layers 1-2) tabulated per step on 2 048 (finite "
                    "scale) / 16 384 (all-pairs scale) length intervals by the edge tile's own code, 4-point Lagrange interpolation per edge; the "
                    "generator's time is inside avg_launch_ms; deviation from the per-edge evaluation <= 4e-6 of the score, both 3e-6..2e-5 from "
                    "the fp64 oracle (tests/test_gpu_parity.py::test_radial_table_of_the_sampler_against_the_per_edge_evaluation); "
                    "--no-radial-table measures the per-edge path"}


def extractor_times(n_scene: int, n_grasp: int, device, reps: int = 5, lmax: int = 2):
    import numpy as np
    from diffusion_edf_amd import synthetic
    from diffusion_edf_amd.gnn_data import FeaturedPoints
    from diffusion_edf_amd.keypoint_extractor import KeypointExtractor
    from diffusion_edf_amd.unet import UnetFeatureExtractor

    def cloud(x):
        x = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(device)
        return FeaturedPoints(x=x, f=torch.rand(len(x), 3, device=device), b=torch.zeros(len(x), dtype=torch.long, device=device), w=None)

    def timed(fn):
        fn(); fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            out = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3, out

    unet = UnetFeatureExtractor(**synthetic.unet_kwargs("panda_lowres" + ("_lmax3" if lmax == 3 else "")), deterministic=True).to(device)
    scene = cloud(synthetic.make_scene(n_scene, seed=0))
    ms_u, levels = timed(lambda: unet(scene))
    if lmax == 3:          # (BASELINE config 5 names the UNet; the keypoint extractor's fields are measured at the shipped lmax 2)
        return {"unet_scene": ms_u, "unet_levels": [len(l.x) for l in levels], "unet_irreps_output": "64x0e+32x1e+16x2e+8x3e"}
    kp = KeypointExtractor(**synthetic.keypoint_extractor_kwargs(bbox=None), deterministic=True).to(device)
    grasp = cloud(synthetic.make_grasp(n_grasp, seed=0))
    ms_k, q = timed(lambda: kp(grasp))
    return {"unet_scene": ms_u, "unet_levels": [len(l.x) for l in levels], "keypoint_grasp": ms_k, "keypoints": len(q.x)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--poses-per-gpu", type=int, default=1000)
    ap.add_argument("--lmax", type=int, default=2)
    ap.add_argument("--scene", type=int, default=4096)
    ap.add_argument("--grasp", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extractors", action="store_true", help="skip the (untimed) feature-extractor measurement")
    ap.add_argument("--no-small-batches", action="store_true", help="skip the (untimed) 16 / 64 / 256-pose rows: profiles/collect.sh, so that per-kernel averages of a trace are those of the timed launches")
    ap.add_argument("--no-score-fwd", action="store_true", help="skip the (untimed) score-forward figure at t = 0.5: profiles/collect.sh, so that a kernel trace holds the sampler's launches only")
    ap.add_argument("--no-radial-table", action="store_true", help="evaluate the radial network's front per edge in the sampler too (A/B; the default tabulates it per step)")
    ap.add_argument("--config5", action="store_true", help="BASELINE config 5 as one assembled model: lmax 3, 16 384-point scene -> UNet key model -> key clouds 3277/656/132/27, "
                                                          "1 024-point grasp -> KeypointExtractor -> query EDF, then the sampler; NOT the headline configuration (C2)")
    ap.add_argument("--drift", action="store_true", help="config 5 only: the Langevin step size of the other configs (0.04) instead of holding the poses at their seeded positions")
    ap.add_argument("--half", action="store_true", help="half-precision GEMM mode (model.half(), the reference's half_precision knob); NOT the headline configuration")
    args = ap.parse_args()
    if args.config5:
        args.lmax, args.scene = 3, 16384

    import torch.distributed as dist
    from diffusion_edf_amd import dist as ddist
    from diffusion_edf_amd.score_head import ScoreModelHead
    from diffusion_edf_amd.score_model_base import ScoreModelBase

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product has no CPU path")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    use_dist = world > 1 or "RANK" in os.environ          # under torch.distributed.run always go through RCCL
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=device)

    n_total = args.poses_per_gpu * world
    first, _ = ddist.shard_range(n_total, world, rank)
    extract5 = None
    if args.config5:
        from diffusion_edf_amd import synthetic
        model, keys, query, extract5 = build_config5(args.scene, args.grasp, device)
        head = model.score_head
        Ts = synthetic.make_poses(args.poses_per_gpu, seed=1, first_pose_index=first).to(device)
        if args.half:
            head.half()
    else:
        kw, cfg, P, keys, query, Ts = build_inputs(args.lmax, args.scene, args.grasp, args.poses_per_gpu, first, device)
        head = ScoreModelHead(**kw)
        head.load_state_dict(P)
        head.to(device)
        if args.half:
            head.half()
        model = ScoreModelBase(head)
    if args.no_radial_table:
        head.set_radial_table(False)
    head.set_key_clouds(keys)
    head.set_query(query)

    # Config 5 holds the poses where they were seeded (uniform in the workspace: SURVEY 8(d), ~55-95 edges per node on the 16 k-point scene): the
    # full step runs -- score, noise, SE(3) update -- with a step size of 1e-6, so that the line measures the 16 k-point / lmax-3 cost per step and
    # not where a random-init network happens to push the poses (with dt = 0.04 they drift into the dense plane within a few steps: 15-23 M edges
    # per step, +-30 % between runs; --drift restores that).
    dt = 1e-6 if (args.config5 and not args.drift) else 0.04

    def run(T0, n_steps, first_idx):
        # t: 1 -> 0.15 log-spaced, dt 0.04, temperature 1 (configs C1/C2 of BASELINE.md)
        return model.sample(T0, keys, query, [[1.0, 0.15]], [n_steps], [dt], temperatures=1.0, seed=3, first_pose_index=first_idx)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    T = Ts
    if args.warmup > 0:
        T = run(T, args.warmup, first)[-1]
        if use_dist:
            ddist.gather_poses(T, n_total)
    head.profile_enable(True)
    head.profile_read()
    barrier()
    t0 = time.perf_counter()
    traj = run(T, args.steps, first)
    final = ddist.gather_poses(traj[-1], n_total) if use_dist else traj[-1]
    barrier()
    el = time.perf_counter() - t0
    prof = head.profile_read()
    head.profile_enable(False)
    if use_dist:
        tt = torch.tensor([el], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())
    assert torch.isfinite(final).all()

    # The same K steps from the same poses with the sampler's radial table switched OFF (every edge evaluates the whole radial network), timed
    # the same way on every rank: reported next to `value` so that both readings of the workload are in the one JSON line.
    per_edge = None
    if args.lmax in (2, 3) and not args.half and not args.no_radial_table:
        head.set_radial_table(False)
        run(T, 2, first)
        head.profile_enable(True)
        head.profile_read()
        barrier()
        t1 = time.perf_counter()
        traj2 = run(T, args.steps, first)
        if use_dist:
            ddist.gather_poses(traj2[-1], n_total)
        barrier()
        el2 = time.perf_counter() - t1
        prof2 = head.profile_read()
        head.profile_enable(False)
        head.set_radial_table(True)
        if use_dist:
            tt = torch.tensor([el2], device=device, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el2 = float(tt.item())
        per_edge = {"value": n_total * args.steps / el2, "ms_per_step": el2 / args.steps * 1e3,
                    "k_edge_ms": prof2["ms"]["edge"] / max(1, prof2["n_evals"]),
                    # after ONE step from the same poses with the same noise (later steps diverge: the Langevin map of a random-init
                    # network is chaotic, any rounding difference grows by orders of magnitude over K steps)
                    "max_pose_difference_after_1_step": float((traj2[1] - traj[1]).abs().max()),
                    "pose_displacement_of_that_step": float((traj[1] - traj[0]).abs().max())}

    # outside the timed region: the score forward alone on the SEEDED poses at the fixed time t = 0.5 --
    # a figure that does not depend on --steps (the edge count of a trajectory drifts with the number of steps taken)
    fixed = None
    if rank == 0 and not args.no_score_fwd:
        Tf = Ts.float()
        tf = torch.full((len(Tf),), 0.5, device=device)
        for _ in range(2):
            head(Tf, keys, query, tf)
        head.profile_enable(True)
        head.profile_read()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n_rep = 10
        e0.record()
        for _ in range(n_rep):
            head(Tf, keys, query, tf)
        e1.record()
        torch.cuda.synchronize()
        pf = head.profile_read()
        head.profile_enable(False)
        fixed = {"ms": e0.elapsed_time(e1) / n_rep, "edges": pf["n_edges"] / max(1, pf["n_evals"]),
                 "k_edge_ms": pf["ms"]["edge"] / max(1, pf["n_evals"])}

    # outside the timed region too: the step BEFORE the path on clouds of the workload's sizes -- the UNet key model on
    # the scene cloud and the KeypointExtractor query model on the grasp cloud, once per agent.sample in a deployment
    extract = None
    if extract5 is not None:
        extract = extract5
    elif rank == 0 and world == 1 and args.lmax in (2, 3) and not args.no_extractors:
        extract = extractor_times(args.scene, args.grasp, device, lmax=args.lmax)

    # outside the timed region as well: the deployment regime of the reference (N_samples = 10 / 20 poses per call, evaluate_real_mug.ipynb:188-190)
    # -- the same scene, the first 16 / 64 / 256 of the seeded poses, 50 steps of the same schedule; latency-bound (one round of edge tiles)
    small = None
    if rank == 0 and world == 1 and not args.half and not args.no_small_batches:
        small = {}
        for n_small in (16, 64, 256):
            if n_small >= args.poses_per_gpu:
                continue
            run(Ts[:n_small], 5, 0)
            torch.cuda.synchronize()
            reps = []
            for _ in range(7):          # 7 repetitions of the 50-step call; the median is the figure, the spread is reported beside it
                ts = time.perf_counter()
                run(Ts[:n_small], 50, 0)
                torch.cuda.synchronize()
                reps.append((time.perf_counter() - ts) / 50 * 1e3)
            reps.sort()
            med = reps[len(reps) // 2]
            small[f"{n_small} poses"] = {"ms_per_step": med, "pose_steps_per_s": n_small / med * 1e3, "min_ms": reps[0], "max_ms": reps[-1], "repetitions": len(reps)}
        # The reference's own deployment shape (round-5 review, item 3): pick models run 10-20 poses x the TWO static keypoints of pick_lowres
        # (configs/panda_mug/pick_lowres/score_model_configs.yaml:76-80, evaluate_real_mug.ipynb:188-190) in 200-step calls (configs/panda_mug/server.yaml:2);
        # place models 20 poses x the grasp's ~100 query points.  Same scene, the first 20 seeded poses, median of 5 calls of 200 steps.
        from diffusion_edf_amd.gnn_data import FeaturedPoints as _FP
        gq = torch.Generator().manual_seed(5)
        q2 = _FP(x=torch.tensor([[0.5, 0.5, 10.5], [-0.5, -0.5, 10.5]], device=device), f=torch.randn(2, query.f.shape[1], generator=gq).to(device),
                 b=torch.zeros(2, dtype=torch.long, device=device), w=torch.sigmoid(torch.randn(2, generator=gq)).to(device))
        for label, q in (("20 poses x 2 static keypoints, 200-step calls", q2), (f"20 poses x {len(query.x)} query points, 200-step calls", query)):
            call = lambda ns, q=q: model.sample(Ts[:20], keys, q, [[1.0, 0.15]], [ns], [dt], temperatures=1.0, seed=3, first_pose_index=0)
            call(5)
            torch.cuda.synchronize()
            reps = []
            for _ in range(5):
                ts = time.perf_counter()
                call(200)
                torch.cuda.synchronize()
                reps.append((time.perf_counter() - ts) / 200 * 1e3)
            reps.sort()
            med = reps[len(reps) // 2]
            small[label] = {"ms_per_step": med, "pose_steps_per_s": 20 / med * 1e3, "min_ms": reps[0], "max_ms": reps[-1], "repetitions": len(reps)}
        head.set_query(query)

    if rank == 0:
        n_ev = max(1, prof["n_evals"])
        edge_ms = prof["ms"]["edge"] / n_ev
        e_per_launch = prof["n_edges"] / n_ev
        flops = 2.0 * e_per_launch * M_EDGE[args.lmax]
        achieved = flops / (edge_ms * 1e-3) / 1e12 if edge_ms > 0 else 0.0
        peak = PEAK_FP16_MFMA_TFLOPS / (1.0 if args.half else 3.0)
        default_workload = (args.lmax, args.scene, args.grasp, args.poses_per_gpu) == (2, 4096, 1024, 1000)
        default_workload = default_workload and not args.half
        wname = "config 5 (one assembled model: UNet key model + KeypointExtractor query model + lmax-3 score head)" if args.config5 else "C2" if default_workload else ("C1" if (args.lmax, args.scene, args.grasp) == (1, 2048, 512) else
                                                ("C2 inputs at lmax 3 (the degree BASELINE config 5 names)" if (args.lmax, args.scene, args.grasp) == (3, 4096, 1024) else "custom"))
        traffic, traffic_src, mfma_issued = None, None, None
        # (files are named per round, r01i < r02h < r03i ...: the last one in name order that holds the HBM passes of the headline kernel wins)
        lmax3_workload = (args.lmax, args.scene, args.grasp, args.poses_per_gpu) == (3, 4096, 1024, 1000) and not args.half and not args.no_radial_table
        # (tags: rNN + one or two letters)
        pmc_re = r"r\d\d[a-z]{1,2}_pmc_summary\.json" if default_workload else (r"r\d\d[a-z]{1,2}_lmax3_pmc_summary\.json" if lmax3_workload else None)
        if args.config5:
            pmc_re = r"r\d\d[a-z]{1,2}_config5_pmc_summary\.json" if (args.poses_per_gpu == 1000 and not args.half and not args.no_radial_table) else None
        import re as _re
        for f in sorted(g for g in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")) if _re.fullmatch(pmc_re, os.path.basename(g))) if pmc_re else []:
            try:
                doc = json.load(open(f))
                v = doc.get("edge_kernel_hbm_bytes_per_launch")
                if v is not None:
                    traffic, traffic_src = v, "profiles/" + os.path.basename(f) + " (separate rocprofv3 --pmc passes of the bench command profiles/collect.sh names)"
                    mfma_issued = doc.get("edge_kernel_frac_mfma_issued")
            except Exception:
                pass
        # FLOP the edge kernel EXECUTES per edge: with the sampler's radial table the front of the radial network (40 960 MAC) runs per grid
        # node, not per edge.  `frac` is on the executed count (a hardware utilisation figure); `frac_algorithmic` keeps the reference's
        # algorithmic count of SURVEY 8(d) (what the work is worth, not what the pipes did).
        table_on = args.lmax in (2, 3) and not args.half and not args.no_radial_table
        m_exec = M_EDGE[args.lmax] - (M_EDGE_FRONT[args.lmax] if table_on else 0)
        # Round 5: the edge-aligned-frame kernels (DEDF_SO2 != 0, full precision) execute fewer multiply-adds than that for the same result -- per
        # path one product per output component instead of the Clebsch-Gordan sum, and no GEMM terms for components a path cannot reach.  `frac`
        # stays on the SAME constant as rounds 2-4 (SURVEY 8(d)'s algorithmic work minus what the table serves): it is the edge rate in fixed
        # units, comparable across rounds; what the pipes really did is `gemm_mac_per_edge_executed` / `frac_gemm_executed` and `frac_mfma_issued`.
        so2_on = os.environ.get("DEDF_SO2", "1") != "0" or args.half      # (the half-precision mode has the edge-frame kernels only)
        gemm_general = M_EDGE[args.lmax] - M_EDGE_CG[args.lmax] - (M_EDGE_FRONT[args.lmax] if table_on else 0)      # (true shapes, as SURVEY counts them)
        gemm_exec = edge_frame_gemm_mac(args.lmax, table_on) if so2_on else gemm_general
        flops_exec = 2.0 * e_per_launch * m_exec
        achieved_exec = flops_exec / (edge_ms * 1e-3) / 1e12 if edge_ms > 0 else 0.0
        out = {
            "metric": f"denoised SE(3) poses/sec (pose-steps/s, {args.scene // 1024}k-pt scene / {args.grasp}-pt grasp, lmax={args.lmax})",
            "value": n_total * args.steps / el,
            "unit": "pose-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": el / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ("f16 GEMM operands (half-precision mode: one fp16 MFMA product per GEMM, fp32 accumulate; rest f32) + f64 (SE(3) Langevin state)" if args.half else
                      "f32 (score network; dense GEMMs as 3-term split-fp16 MFMA products = 22-bit operands, fp32 accumulate) + f64 (SE(3) Langevin state)"),
            "data": "synthetic (seeded scene/grasp clouds of the named sizes, random-init weights of the reference architecture)",
            "config": {"workload": f"{wname}: {args.scene}-pt scene -> key clouds {'/'.join(str(len(k.x)) for k in keys)}, {args.grasp}-pt grasp -> "
                                   f"{len(query.x)} query pts, lmax={args.lmax}, {args.poses_per_gpu} poses per GPU, t 1->0.15 log-spaced, dt 0.04",
                       "poses_total": n_total, "parallelism": f"pose-parallel dp{world}, one RCCL all-gather at the end",
                       "edges_per_step_rank0": e_per_launch, "full_trajectories_per_s_at_50_steps": n_total * args.steps / el / 50.0,
                       "score_fwd_ms_at_t0.5": fixed, "feature_extractors_ms": extract, "small_batches_50_steps": small},
            "roofline": {"kernel": "k_edge (fused per-edge pipeline)", "bound": "mfma", "limiter": "valu-issue (one wave per SIMD: VALU and MFMA time add up, DESIGN.md section 5.R5)",
                         "achieved": achieved, "peak": peak,
                         "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic, "traffic_source": traffic_src,
                         "frac_definition": f"2 x {M_EDGE[args.lmax]} MAC per edge (SURVEY 8(d)'s algorithmic constant, dense-CG convention) x edges of the launch / the kernel's HIP-event duration / peak",
                         "frac_round2_4_units": achieved_exec / peak,
                         "frac_round2_4_units_definition": f"the same edge rate priced at 2 x {m_exec} MAC per edge (what the GENERAL form of the kernel executed in rounds 2-4" + (": the algorithmic count minus the 40 960 MAC the sampler's radial table evaluates per grid node" if table_on else "") + "); kept so that the rounds stay comparable, not a utilisation figure",
                         "edge_frame_kernels": so2_on,
                         "gemm_mac_per_edge_executed": gemm_exec, "gemm_mac_per_edge_general_form": gemm_general,
                         "frac_gemm_executed": (2.0 * e_per_launch * gemm_exec / (edge_ms * 1e-3) / 1e12 / peak) if edge_ms > 0 else 0.0,
                         "frac_gemm_executed_definition": "2 x the dense-GEMM multiply-adds the launched kernels really execute per edge (kernel shapes, padding included; the lane-local VALU work -- rotations, operand forming, activations, softmax partials -- not counted) x edges / launch time / peak",
                         "frac_mfma_issued": mfma_issued, "frac_mfma_issued_definition": "SQ_INSTS_MFMA x 32 768 FLOP / launch time in the kernel trace / 2.5 PFLOP/s, from the file `traffic_source` names (the MFMAs the kernel issues, all three split terms counted)",
                         "peak_definition": "dense fp16 MFMA peak 2500 TFLOP/s / 3 (every GEMM is a 3-term split-fp16 product, fp32 accumulate): the hardware matrix peak of the arithmetic the kernel uses, in fp32-equivalent FLOP/s",
                         "dtype": "f16 x3 split (22-bit operands), f32 accumulate",
                         "mix_bound": {"peak": mix_peak_tflops(args.lmax), "frac": achieved / mix_peak_tflops(args.lmax),
                                       "definition": "GEMMs on the split-fp16 MFMA peak in series with the dense-convention Clebsch-Gordan MACs on the 157.3 TFLOP/s fp32 VALU peak"},
                         "avg_launch_ms": edge_ms, "algorithmic_flop_per_launch": flops,
                         "radial_table": radial_table_note(args, flops, e_per_launch, edge_ms, peak, per_edge),
                         "kernel_ms_per_step": {k: v / n_ev for k, v in prof["ms"].items()}},
        }
        if not args.no_cpu_baseline and world == 1:           # the CPU reference leg is timed at N = 1 only
            # (config 5: same key / query cloud SIZES and positions from the synthetic generator, random features -- the CPU leg's time does not depend on
            #  the feature values; a smaller pose sample, ~95 edges per node at this scene)
            out["cpu_baseline"] = cpu_baseline(args.lmax, args.scene, args.grasp, n_sample_poses=8 if args.scene > 8192 else 32)
        print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
