"""Graph construction of the feature extractors — mirrors reference ``diffusion_edf/connectivity.py`` (``RadiusGraph`` :8-30,
``RadiusConnect`` :34-49, ``FpsPool`` :53-80) on the HIP primitives ``dedf_fps`` / ``dedf_radius`` instead of torch_cluster /
torch_scatter.  Single cloud (every shipped config has all batch indices 0); inputs must live on the GPU — there is no CPU path.
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
    assert x.ndim == 2 and x.shape[-1] == 3, f"{tuple(x.shape)}"
    if not x.is_cuda:
        raise RuntimeError("diffusion_edf_amd.connectivity needs GPU tensors: the product has no CPU path")
    if batch is not None and batch.numel() and int(batch.max()) != int(batch.min()):
        raise NotImplementedError("several clouds in one batch vector (every shipped config uses a single cloud)")


def fps(src: torch.Tensor, batch: Optional[torch.Tensor] = None, ratio: float = 0.5, random_start: bool = True, _trusted: bool = False) -> torch.Tensor:
    """torch_cluster.fps as connectivity.py:62 calls it: ``ceil(ratio * N)`` indices in selection order.
    (``_trusted``: the caller has already checked that the batch vector holds one cloud — the check reads back from the device.)"""
    if not _trusted:
        _check_cloud(src, batch)
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
        _check_cloud(x, batch_x)
        _check_cloud(y, batch_y)
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
    if not _trusted:
        _check_cloud(x, batch)
    return radius(x, x, r, batch, batch, max_num_neighbors, _exclude_self=not loop, _trusted=True)


def _in_degree(edge_dst: torch.Tensor, n_nodes: int) -> torch.Tensor:
    """number of incoming edges per destination node (the reference's scatter_add of ones)"""
    return torch.bincount(edge_dst, minlength=n_nodes).to(edge_dst.dtype)


class RadiusGraph(torch.nn.Module):
    """Self graph of one cloud: every node is connected to its neighbours within ``r`` (no self loops).
    Returns ``(features, coordinates, edge_src, edge_dst, degree, batch)`` like reference connectivity.py:14-29."""

    def __init__(self, r: float, max_num_neighbors: int):
        super().__init__()
        self.r, self.max_num_neighbors = r, max_num_neighbors

    def forward(self, node_coord_src: torch.Tensor, node_feature_src: torch.Tensor, batch_src: torch.Tensor, _trusted: bool = False):
        dst, src = radius_graph(node_coord_src, self.r, batch_src, loop=False, max_num_neighbors=self.max_num_neighbors, _trusted=_trusted)
        return node_feature_src, node_coord_src, src, dst, _in_degree(dst, len(node_coord_src)), batch_src


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

    def forward(self, node_coord_src: torch.Tensor, node_feature_src: torch.Tensor, batch_src: torch.Tensor, _trusted: bool = False):
        picked = fps(node_coord_src, batch_src, ratio=self.ratio, random_start=self.random_start, _trusted=_trusted)
        coord, feat, batch = node_coord_src[picked], node_feature_src[picked], batch_src[picked]
        src, dst = self.radius_connect(node_coord_src, batch_src, coord, batch, _trusted=True)      # (a sub-sample of a single cloud is one)
        other = picked[dst] != src                      # drop the edge from a pooled node to the source point it was sampled from
        src, dst = src[other], dst[other]
        return feat, coord, src, dst, _in_degree(dst, len(picked)), batch
