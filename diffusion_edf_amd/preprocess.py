"""Point-cloud pre-processing named by the reference's ``preprocess.yaml`` files (``configs/*/preprocess.yaml``:
``[crop_bbox →] downsample → rescale``; the two sapien task files crop the scene cloud, the panda files have the crop commented out).

The reference composes them from ``edf_interface.data.preprocess`` (``train_utils.py:24-31``), an un-vendored submodule that
is absent from the reference tree — semantics restated from the YAML comments (metres → centimetres, voxel size in metres,
``coord_reduction: average``), "parity unpinned".  Here they are plain functions over the boundary types of this build:

* ``FeaturedPoints``  — coordinates (and, for ``downsample``, features) are processed, batch / weights follow;
* ``(n, 7)`` pose tensors ``[qw,qx,qy,qz,x,y,z]`` — ``rescale`` multiplies the translations, the other procs leave poses alone;
* anything else passes through unchanged (the agent hands its three inputs — scene, grasp, poses — to the same function,
  reference ``agent.py:126-128``).

Runs on whatever device the inputs live on (torch ops; called once per ``agent.sample``, not on the per-step hot path).
"""
from __future__ import annotations

from functools import partial
from typing import Any, Callable, Dict, List, Optional, Sequence

import torch

from .gnn_data import FeaturedPoints


def _is_poses(obj) -> bool:
    return isinstance(obj, torch.Tensor) and obj.ndim == 2 and obj.shape[-1] == 7


def rescale(obj, rescale_factor: float):
    """lengths × factor: point coordinates, pose translations"""
    if isinstance(obj, FeaturedPoints):
        return obj._replace(x=obj.x * rescale_factor)
    if _is_poses(obj):
        out = obj.clone()
        out[:, 4:] = out[:, 4:] * rescale_factor
        return out
    return obj


def crop_bbox(obj, bbox: Sequence[Sequence[float]], targets: Optional[Sequence[str]] = None, role: Optional[str] = None):
    """keep the points inside ``[[x0,x1],[y0,y1],[z0,z1]]``.  ``targets`` names the inputs the crop applies to (the sapien task
    files crop ``['scene_pcd']`` only); ``role`` is what the caller says the object is (``'scene_pcd'``, ``'grasp_pcd'``,
    ``'poses'``) -- with ``targets`` given and a role outside it, the object passes through.  With ``targets`` given and NO role the call is
    ambiguous (a grasp cloud cropped to the scene's box comes back empty): it raises instead of guessing."""
    if not isinstance(obj, FeaturedPoints):
        return obj
    if targets is not None:
        if role is None:
            raise ValueError(f"crop_bbox: this proc applies to {list(targets)} only -- pass role='scene_pcd' | 'grasp_pcd' (DiffusionEdfAgent.sample does)")
        if role not in targets:
            return obj
    lo = torch.tensor([b[0] for b in bbox], dtype=obj.x.dtype, device=obj.x.device)
    hi = torch.tensor([b[1] for b in bbox], dtype=obj.x.dtype, device=obj.x.device)
    keep = ((obj.x >= lo) & (obj.x <= hi)).all(dim=-1)
    return FeaturedPoints(x=obj.x[keep], f=obj.f[keep], b=obj.b[keep], w=None if obj.w is None else obj.w[keep])


def downsample(obj, voxel_size: float, coord_reduction: str = "average"):
    """one point per occupied voxel (per batch index): coordinates ``average`` (mean of the voxel's points) or ``center`` (voxel
    centre), features averaged, weights averaged.  Output order: ascending (batch, voxel ix, iy, iz)."""
    if not isinstance(obj, FeaturedPoints):
        return obj
    if coord_reduction not in ("average", "center"):
        raise ValueError(f"Unknown coord_reduction: {coord_reduction}")
    if len(obj.x) == 0:
        return obj
    ijk = torch.floor(obj.x / voxel_size).to(torch.int64)
    key = torch.cat([obj.b.to(torch.int64)[:, None], ijk], dim=-1)
    uniq, inv = torch.unique(key, dim=0, return_inverse=True)          # sorted lexicographically
    # voxel means in a FIXED summation order (points of a voxel in input order, one sequential pass per voxel): scatter-adds with atomics
    # would make the coordinates -- and with them every graph and score downstream -- differ in the last bits from call to call
    order = torch.sort(inv, stable=True).indices
    counts = torch.bincount(inv, minlength=len(uniq))

    def mean(v):
        return torch.segment_reduce(v.index_select(0, order), "mean", lengths=counts, axis=0)

    x = mean(obj.x) if coord_reduction == "average" else (uniq[:, 1:].to(obj.x.dtype) + 0.5) * voxel_size
    return FeaturedPoints(x=x, f=mean(obj.f), b=uniq[:, 0].to(obj.b.dtype), w=None if obj.w is None else mean(obj.w))


PROCS: Dict[str, Callable] = {"rescale": rescale, "crop_bbox": crop_bbox, "downsample": downsample}


def compose_proc_fn(preprocess_config: Optional[List[Dict[str, Any]]], registry: Optional[Dict[str, Callable]] = None) -> Callable:
    """``[{name, kwargs}, ...]`` (the ``preprocess_config`` / ``unprocess_config`` lists of preprocess.yaml) → one callable that
    applies the procs in order.  Unknown names raise ``AttributeError`` like the reference's ``getattr(preprocess, name)``."""
    reg = dict(PROCS)
    reg.update(registry or {})
    import inspect
    steps = []
    for proc in preprocess_config or []:
        if proc["name"] not in reg:
            raise AttributeError(f"unknown preprocess step '{proc['name']}' (known: {sorted(reg)})")
        fn = reg[proc["name"]]
        steps.append((partial(fn, **(proc.get("kwargs") or {})), "role" in inspect.signature(fn).parameters))

    def proc_fn(obj, role: Optional[str] = None):
        for step, takes_role in steps:
            obj = step(obj, role=role) if takes_role else step(obj)
        return obj

    proc_fn.accepts_role = True
    return proc_fn
