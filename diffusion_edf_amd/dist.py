"""Pose-parallel data parallelism: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over xGMI on ROCm;
"gloo" in the CPU tests).

The reference has no distributed code at all.  Poses are independent units: every pose's denoising
trajectory depends only on its own ``T``, the shared read-only key/query clouds and the weights, so the path shards with
no per-step communication.  Ranks take contiguous blocks of the pose batch (remainder to the low ranks), every rank holds
the (few-MB) clouds and weights, noise is keyed by the *global* pose index, and ONE all-gather of the final poses (or of
the whole trajectory) closes the run.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n_total: int, world_size: int, rank: int) -> Tuple[int, int]:
    """contiguous block [start, stop) of rank `rank`; remainder goes to the low ranks"""
    base, rem = divmod(n_total, world_size)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def shard_poses(T: torch.Tensor, world_size: Optional[int] = None, rank: Optional[int] = None) -> Tuple[torch.Tensor, int]:
    """-> (local block of poses, global index of its first pose)"""
    ws = dist.get_world_size() if world_size is None else world_size
    rk = dist.get_rank() if rank is None else rank
    s, e = shard_range(len(T), ws, rk)
    return T[s:e], s


def gather_poses(local: torch.Tensor, n_total: int, dim: int = 0, group=None) -> torch.Tensor:
    """All-gather pose blocks of (possibly) unequal length along `dim` into the full batch, identical on every rank.
    Uses one all_gather on blocks padded to the largest shard (a single RCCL collective)."""
    ws = dist.get_world_size(group)
    sizes = [shard_range(n_total, ws, r) for r in range(ws)]
    mx = max(e - s for s, e in sizes)
    x = local.movedim(dim, 0).contiguous()
    if mx == 0:                       # no poses at all: every rank knows it (n_total is global), no collective needed
        return local
    pad = torch.zeros((mx,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    pad[: x.shape[0]] = x
    out = [torch.empty_like(pad) for _ in range(ws)]
    dist.all_gather(out, pad, group=group)
    full = torch.cat([o[: e - s] for o, (s, e) in zip(out, sizes)], dim=0)
    return full.movedim(0, dim)


def sample_sharded(model, T_seed: torch.Tensor, scene_pcd_multiscale, grasp_pcd, *, gather_trajectory: bool = False,
                   seed: int = 0, **sample_kwargs) -> torch.Tensor:
    """``ScoreModelBase.sample`` over the pose shard of this rank + the closing all-gather.
    Returns the final poses (nT,7) — or the whole (steps+2, nT, 7) trajectory — of ALL poses on every rank."""
    nT = len(T_seed)
    local, first = shard_poses(T_seed)          # may be empty (world size > number of poses): sample() then returns an empty block
    traj = model.sample(local, scene_pcd_multiscale, grasp_pcd, seed=seed, first_pose_index=first, **sample_kwargs)
    if gather_trajectory:
        return gather_poses(traj, nT, dim=1)
    return gather_poses(traj[-1], nT, dim=0)
