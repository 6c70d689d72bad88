"""Plain tensor containers of the hot-path boundary — same fields and order as reference
diffusion_edf/gnn_data.py:12-16 (FeaturedPoints) so callers can pass the reference's own NamedTuples."""
from typing import NamedTuple, Optional

import torch


class FeaturedPoints(NamedTuple):
    x: torch.Tensor                     # (N, 3) positions [cm]
    f: torch.Tensor                     # (N, F) irreps features, blocks 0e|1e|2e, mul-major, m fastest
    b: torch.Tensor                     # (N,) batch index (always 0 on this path)
    w: Optional[torch.Tensor] = None    # (N,) optional point weights


def flatten_featured_points(points: FeaturedPoints) -> FeaturedPoints:
    """reference gnn_data.py:103-113"""
    w = points.w.reshape(-1) if points.w is not None else None
    return FeaturedPoints(x=points.x.reshape(-1, 3), f=points.f.reshape(-1, points.f.shape[-1]),
                          b=points.b.reshape(-1), w=w)
