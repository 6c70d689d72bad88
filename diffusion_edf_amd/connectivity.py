"""Graph construction of the feature extractors — mirrors reference ``diffusion_edf/connectivity.py`` (``RadiusGraph`` :8-30,
``RadiusConnect`` :34-49, ``FpsPool`` :53-80) on the HIP primitives ``dedf_fps`` / ``dedf_radius`` instead of torch_cluster /
torch_scatter.  Inputs must live on the GPU — there is no CPU path.  Several clouds in one batch vector (``batch`` sorted, as torch_cluster
requires; reference connectivity.py:62 passes ``batch`` to ``fps``, :43 ``batch_x`` / ``batch_y`` to ``radius``) are handled cloud by cloud on
the same primitives; every shipped config has all batch indices 0, which is the fast path (no host round trip per graph).
The UNet / keypoint extractors that consume these graphs: ``unet.py``, ``keypoint_extractor.py``."""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional, Tuple

import torch

from . import _lib


def _stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _check_cloud(x: torch.Tensor, batch: Optional[torch.Tensor]):
    """-> None for a single cloud, else [(batch id, start, end)] of the clouds of a SORTED batch vector (reads back from the device)"""
    assert x.ndim == 2 and x.shape[-1] == 3, f"{tuple(x.shape)}"
    if not x.is_cuda:
        raise RuntimeError("diffusion_edf_amd.connectivity needs GPU tensors: the product has no CPU path")
    if batch is None or batch.numel() == 0:
        return None
    assert batch.ndim == 1 and len(batch) == len(x), f"{tuple(batch.shape)}"
    ids, counts = torch.unique_consecutive(batch, return_counts=True)
    if len(ids) == 1:
        return None
    ids, counts = ids.tolist(), counts.tolist()
    if any(b <= a for a, b in zip(ids, ids[1:])):
        raise ValueError("batch vector must be sorted (torch_cluster's requirement)")
    segs, s0 = [], 0
    for b, c in zip(ids, counts):
        segs.append((b, s0, s0 + c))
        s0 += c
    return segs


def fps(src: torch.Tensor, batch: Optional[torch.Tensor] = None, ratio: float = 0.5, random_start: bool = True, _trusted: bool = False) -> torch.Tensor:
    """torch_cluster.fps as connectivity.py:62 calls it: ``ceil(ratio * N)`` indices in selection order.
    (``_trusted``: the caller has already checked that the batch vector holds one cloud — the check reads back from the device.)"""
    segs = None if _trusted else _check_cloud(src, batch)
    if segs is not None:              # cloud by cloud: ceil(ratio * n_b) points of each, indices into the concatenated cloud
        return torch.cat([fps(src[a:b], None, ratio=ratio, random_start=random_start, _trusted=True) + a for _, a, b in segs])
    n = len(src)
    k = int(math.ceil(ratio * n))
    start = int(torch.randint(n, (1,)).item()) if random_start else 0
    x = src.detach().to(torch.float32).contiguous()
    idx = torch.empty(k, device=src.device, dtype=torch.int32)
    lib = _lib.load()
    with torch.cuda.device(src.device):          # dedf_fps / dedf_radius run on the CURRENT device and stream
        rc = lib.dedf_fps(x.data_ptr(), n, k, start, idx.data_ptr(), _stream())
    if rc != _lib.OK:
        raise (NotImplementedError if rc == _lib.ERR_UNSUPPORTED else RuntimeError)(f"dedf_fps failed ({rc}); clouds up to 65 536 points")
    return idx.long()


def radius(x: torch.Tensor, y: torch.Tensor, r: float, batch_x=None, batch_y=None, max_num_neighbors: int = 32,
           _exclude_self: bool = False, _trusted: bool = False) -> torch.Tensor:
    """torch_cluster.radius: for every point of ``y`` the points of ``x`` within ``r`` -> ``(2, E)`` = [y index, x index]."""
    if not _trusted:
        sx, sy = _check_cloud(x, batch_x), _check_cloud(y, batch_y)
        if sx is not None or sy is not None:      # pairs exist inside a cloud only: cloud by cloud, sorted by y index like the single-cloud result
            if sx is None:
                sx = [(int(batch_x[0]) if batch_x is not None and len(batch_x) else 0, 0, len(x))]
            if sy is None:
                sy = [(int(batch_y[0]) if batch_y is not None and len(batch_y) else 0, 0, len(y))]
            xseg = {b: (a, e) for b, a, e in sx}
            parts = []
            for b, a, e in sy:
                if b not in xseg:
                    continue
                xa, xe = xseg[b]
                edge = radius(x[xa:xe], y[a:e], r, None, None, max_num_neighbors, _exclude_self=_exclude_self, _trusted=True)
                parts.append(torch.stack([edge[0] + a, edge[1] + xa], dim=0))
            return torch.cat(parts, dim=1) if parts else torch.zeros(2, 0, dtype=torch.int64, device=x.device)
    xs = x.detach().to(torch.float32).contiguous()
    ys = y.detach().to(torch.float32).contiguous()
    lib = _lib.load()
    n = C.c_int64(0)
    full = len(ys) * min(int(max_num_neighbors), len(xs))          # the most there can be; taken at once while it is small (<= 256 MB)
    cap = full if full <= (1 << 24) else 32 * len(ys) + 1024
    nbytes = int(lib.dedf_radius_scratch_bytes(len(ys)))
    scratch = torch.empty((nbytes + 7) // 8, device=x.device, dtype=torch.int64)          # caller-owned scratch: the library keeps no state
    while True:
        ed = torch.empty(cap, device=x.device, dtype=torch.int64)
        es = torch.empty(cap, device=x.device, dtype=torch.int64)
        with torch.cuda.device(x.device):
            rc = lib.dedf_radius(xs.data_ptr(), len(xs), ys.data_ptr(), len(ys), float(r), int(max_num_neighbors), int(_exclude_self),
                                 cap, ed.data_ptr(), es.data_ptr(), C.byref(n), scratch.data_ptr(), scratch.numel() * 8, _stream())
        if rc == _lib.OK:
            break
        if rc == _lib.ERR_INVALID and n.value > cap:
            cap = int(n.value)
            continue
        raise RuntimeError(f"dedf_radius failed ({rc})")
    return torch.stack([ed[: n.value], es[: n.value]], dim=0)


def radius_graph(x: torch.Tensor, r: float, batch=None, loop: bool = False, max_num_neighbors: int = 32, _trusted: bool = False) -> torch.Tensor:
    # (loop=False: a point's own index is excluded -- the indices of x and y coincide cloud by cloud as well)
    return radius(x, x, r, batch, batch, max_num_neighbors, _exclude_self=not loop, _trusted=_trusted)


def _in_degree(edge_dst: torch.Tensor, n_nodes: int) -> torch.Tensor:
    """number of incoming edges per destination node (the reference's scatter_add of ones)"""
    return torch.bincount(edge_dst, minlength=n_nodes).to(edge_dst.dtype)


class RadiusGraph(torch.nn.Module):
    """Self graph of one cloud: every node is connected to its neighbours within ``r`` (no self loops).
    Returns ``(features, coordinates, edge_src, edge_dst, degree, batch)`` like reference connectivity.py:14-29."""

    def __init__(self, r: float, max_num_neighbors: int):
        super().__init__()
        self.r, self.max_num_neighbors = r, max_num_neighbors

    def forward(self, node_coord_src: torch.Tensor, node_feature_src: torch.Tensor, batch_src: torch.Tensor, _trusted: bool = False, _need_degree: bool = True):
        dst, src = radius_graph(node_coord_src, self.r, batch_src, loop=False, max_num_neighbors=self.max_num_neighbors, _trusted=_trusted)
        # (`_need_degree=False`: callers that drop the degree -- the UNet's layers count their edges themselves -- skip its device work)
        return node_feature_src, node_coord_src, src, dst, (_in_degree(dst, len(node_coord_src)) if _need_degree else None), batch_src


class RadiusConnect(torch.nn.Module):
    """Bipartite edges from a source cloud to a destination cloud -> ``(edge_src, edge_dst)`` (connectivity.py:34-49)."""

    def __init__(self, r: float, max_num_neighbors: int, offset: Optional[float] = None):
        super().__init__()
        if offset is not None:
            raise NotImplementedError
        self.r, self.max_num_neighbors, self.offset = r, max_num_neighbors, offset

    def forward(self, node_coord_src, batch_src, node_coord_dst, batch_dst, _trusted: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
        dst, src = radius(node_coord_src, node_coord_dst, self.r, batch_src, batch_dst, self.max_num_neighbors, _trusted=_trusted)
        return src, dst


class FpsPool(torch.nn.Module):
    """Pooling by farthest point sampling: the sampled nodes inherit their own features and are connected to the source points
    within ``r``, except to themselves (connectivity.py:53-80).  Same six outputs as ``RadiusGraph``."""

    def __init__(self, ratio: float, random_start: bool, r: float, max_num_neighbors: int):
        super().__init__()
        self.ratio, self.random_start, self.r, self.max_num_neighbors = ratio, random_start, r, max_num_neighbors
        self.radius_connect = RadiusConnect(r=r, max_num_neighbors=max_num_neighbors)

    def forward(self, node_coord_src: torch.Tensor, node_feature_src: torch.Tensor, batch_src: torch.Tensor, _trusted: bool = False,
                _fps_ordered: bool = False, _need_degree: bool = True):
        """``_fps_ordered``: the caller vouches that ``node_coord_src`` is ONE cloud in the selection order of a farthest point sampling that
        started at ITS first point (the previous level of a deterministic FPS cascade).  Sampling such a cloud from its first point again returns its
        own prefix: sample j + 1 of the parent run maximises the distance to samples 1 .. j over the whole parent cloud, hence over the sub-cloud
        too (same distances, bit for bit), and among points tied at that maximum it is the one the parent run took first, i.e. the smallest index
        of the sub-cloud.  ``tests/test_graph.py::test_fps_of_an_fps_ordered_cloud_is_its_prefix`` checks it on the oracle and on the kernels."""
        if _fps_ordered and not self.random_start:
            import math
            import os
            picked = torch.arange(int(math.ceil(self.ratio * len(node_coord_src))), device=node_coord_src.device)
            # The argument above needs every sample of the prefix to have been at NON-ZERO distance from the samples before it when the parent
            # run took it.  Once a run has exhausted the distinct points (clouds with exact duplicates, more samples than distinct points) all
            # remaining distances are 0 and the kernel's smallest-index arg-max returns index 0 again, not the next prefix index: the shortcut
            # does not hold there.  The UNet cascades never get there (ratio 0.2-0.25 of voxel-filtered clouds); DEDF_FPS_CHECK=1 runs the real
            # sampling next to the shortcut and raises on any difference (debugging, tests).
            if os.environ.get("DEDF_FPS_CHECK"):
                real = fps(node_coord_src, batch_src, ratio=self.ratio, random_start=False, _trusted=_trusted)
                if not torch.equal(real.to(picked.device), picked):
                    raise RuntimeError("FpsPool(_fps_ordered=True): the cloud is not in the selection order of a start-0 farthest point sampling "
                                       "with distinct samples; its re-sampling is not its prefix")
        else:
            picked = fps(node_coord_src, batch_src, ratio=self.ratio, random_start=self.random_start, _trusted=_trusted)
        coord, feat, batch = node_coord_src[picked], node_feature_src[picked], batch_src[picked]
        src, dst = self.radius_connect(node_coord_src, batch_src, coord, batch, _trusted=_trusted)      # (a sub-sample of a single cloud is one)
        other = picked[dst] != src                      # drop the edge from a pooled node to the source point it was sampled from
        src, dst = src[other], dst[other]
        return feat, coord, src, dst, (_in_degree(dst, len(picked)) if _need_degree else None), batch
