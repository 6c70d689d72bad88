"""Pose sharding + the closing all-gather, world_size 2 on CPU (gloo).  The per-rank sampler itself needs a GPU; here a
deterministic per-pose function stands in for it, which is exactly what the sharding contract requires: results keyed by
the global pose index must not depend on the partition."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from diffusion_edf_amd import dist as ddist
from diffusion_edf_amd import synthetic


def test_shard_ranges_cover_and_balance():
    for n in (1, 7, 8, 1000, 8001):
        for ws in (1, 2, 3, 8):
            r = [ddist.shard_range(n, ws, k) for k in range(ws)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
            sizes = [e - s for s, e in r]
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)


def test_poses_are_keyed_by_global_index():
    full = synthetic.make_poses(11, seed=1)
    a, b = synthetic.make_poses(4, seed=1), synthetic.make_poses(7, seed=1, first_pose_index=4)
    assert torch.equal(torch.cat([a, b]), full)


class _FakeModel:
    """stands in for ScoreModelBase.sample: trajectory depends only on (global pose index, seed pose)"""

    def sample(self, T, scene, grasp, seed=0, first_pose_index=0, **kw):
        idx = torch.arange(first_pose_index, first_pose_index + len(T), dtype=torch.float64)[:, None]
        steps = [T + s * (idx + 1) * 1e-3 + seed for s in range(3)]
        return torch.stack(steps + [steps[-1]], 0)


def _worker(rank, world, port, n_poses, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    T = synthetic.make_poses(n_poses, seed=1)
    final = ddist.sample_sharded(_FakeModel(), T, None, None, seed=5)
    traj = ddist.sample_sharded(_FakeModel(), T, None, None, seed=5, gather_trajectory=True)
    q.put((rank, final, traj))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_poses", [8, 9, 1])          # 1 pose on 2 ranks: one rank holds an empty shard and must still reach the collective
def test_sharded_run_equals_single_process(n_poses):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_poses, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    T = synthetic.make_poses(n_poses, seed=1)
    ref = _FakeModel().sample(T, None, None, seed=5)
    for rank, final, traj in res:
        assert torch.equal(final, ref[-1]) and torch.equal(traj, ref)


class _FakeStage(_FakeModel):
    """_FakeModel with the extractor hooks and schedules DiffusionEdfAgent.sample uses (agent.py:131-141)"""
    diffusion_schedules = [[1.0, 0.1]]

    def get_key_pcd_multiscale(self, pcd):
        return None

    def get_query_pcd(self, pcd):
        return None

    def sample(self, T_seed=None, scene_pcd_multiscale=None, grasp_pcd=None, seed=0, first_pose_index=0, noise=None, **kw):
        return super().sample(T_seed, scene_pcd_multiscale, grasp_pcd, seed=seed, first_pose_index=first_pose_index)


def _agent_worker(rank, world, port, n_poses, q):
    from diffusion_edf_amd.agent import DiffusionEdfAgent
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    T = synthetic.make_poses(n_poses, seed=1)
    out, _, _ = DiffusionEdfAgent(models=[_FakeStage(), _FakeStage()]).sample(None, None, T, [[3], [3]], [[0.1], [0.1]], [1.0, 1.0], seed=2)
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_agent_cascade_sharded_equals_single_process():
    """two-stage cascade of the agent over 2 ranks: every stage shards the poses and gathers its whole trajectory, so each
    rank returns what a single process returns"""
    from diffusion_edf_amd.agent import DiffusionEdfAgent
    n_poses = 9
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_agent_worker, args=(r, 2, port, n_poses, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    T = synthetic.make_poses(n_poses, seed=1)
    ref, _, _ = DiffusionEdfAgent(models=[_FakeStage(), _FakeStage()]).sample(None, None, T, [[3], [3]], [[0.1], [0.1]], [1.0, 1.0], seed=2)
    assert ref.shape == (8, n_poses, 7)
    for rank, out in res:
        assert torch.equal(out, ref)


# ---- the product under RCCL (one MI355X: world size 1) ----------------------------------------------------------------------------------------

def _nccl_worker(port, q):
    """one rank, backend "nccl" (= RCCL on ROCm): the REAL sampler through dist.sample_sharded -- process-group init on the device, the pose
    shard of the rank, the closing all-gather on the GPU stream -- against the plain single-process call"""
    import sys
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    try:
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import stage_check as SC
        from diffusion_edf_amd.gnn_data import FeaturedPoints
        from diffusion_edf_amd.score_head import ScoreModelHead
        from diffusion_edf_amd.score_model_base import ScoreModelBase
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        kw, cfg, P, keys, query, Ts, _ = SC.build_case(2, 9, 512, 60)
        head = ScoreModelHead(**kw); head.load_state_dict(P); head.to(dev)
        m = ScoreModelBase(head)
        gk = [FeaturedPoints(k.x.to(dev), k.f.to(dev), k.b.to(dev)) for k in keys]
        gq = FeaturedPoints(query.x.to(dev), query.f.to(dev), query.b.to(dev), query.w.to(dev))
        args = dict(diffusion_schedules=[[1.0, 0.5]], N_steps=[3], timesteps=[0.04], temperatures=1.0)
        ref = m.sample(Ts.to(dev), gk, gq, seed=7, **args)
        final = ddist.sample_sharded(m, Ts.to(dev), gk, gq, seed=7, **args)
        traj = ddist.sample_sharded(m, Ts.to(dev), gk, gq, seed=7, gather_trajectory=True, **args)
        dist.barrier()
        ok = bool(torch.equal(final, ref[-1]) and torch.equal(traj, ref) and final.is_cuda and torch.isfinite(traj).all())
        q.put(("ok" if ok else "mismatch", dist.get_backend()))
        dist.destroy_process_group()
    except Exception as e:          # noqa: BLE001
        q.put(("error: " + repr(e), ""))


@pytest.mark.gpu
def test_real_sampler_under_rccl_world_size_one():
    """the product (not a stand-in) under `torch.distributed` with the nccl backend -- RCCL init, shard, sampler, all-gather -- in a spawned rank"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_worker, args=(port, q))
    p.start()
    res, backend = q.get(timeout=600)
    p.join(timeout=120)
    assert res == "ok" and backend == "nccl", (res, backend)
    assert p.exitcode == 0
