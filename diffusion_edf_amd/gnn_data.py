"""Plain tensor containers of the hot-path boundary and their helpers — same names, fields, field order and semantics as reference
``diffusion_edf/gnn_data.py`` (``FeaturedPoints`` :12-16, ``set_featured_points_attribute`` :31-41, ``detach_featured_points``
:44-48, ``merge_featured_points`` :51-74, ``flatten_featured_points`` :103-113, ``GraphEdge`` :117-124,
``set_graph_edge_attribute`` :128-163, ``cat_graph_edges`` :166-218, ``cat_featured_points`` :221-234), so callers can pass the
reference's own NamedTuples either way.  Containers only: no arithmetic happens here.  (``TransformPcd`` :80-100 is done
inside the kernels; ``pcd_to_featured_points`` :77 needs ``edf_interface`` and is out of scope.)"""
from typing import List, NamedTuple, Optional, Sequence, Tuple, Union

import torch


class FeaturedPoints(NamedTuple):
    x: torch.Tensor                     # (N, 3) positions [cm]
    f: torch.Tensor                     # (N, F) irreps features, blocks 0e|1e|2e, mul-major, m fastest
    b: torch.Tensor                     # (N,) batch index (always 0 on this path)
    w: Optional[torch.Tensor] = None    # (N,) optional point weights


class GraphEdge(NamedTuple):
    edge_src: torch.Tensor
    edge_dst: torch.Tensor
    edge_length: Optional[torch.Tensor] = None
    edge_attr: Optional[torch.Tensor] = None
    edge_scalars: Optional[torch.Tensor] = None
    edge_weights: Optional[torch.Tensor] = None
    edge_logits: Optional[torch.Tensor] = None


_KEEP = ''       # the reference's "leave this optional field as it is" marker (a str, because None is a legal new value)


def set_featured_points_attribute(points: FeaturedPoints, x: Optional[torch.Tensor] = None, f: Optional[torch.Tensor] = None,
                                  b: Optional[torch.Tensor] = None, w: Union[str, Optional[torch.Tensor]] = _KEEP) -> FeaturedPoints:
    return FeaturedPoints(x=points.x if x is None else x, f=points.f if f is None else f, b=points.b if b is None else b,
                          w=points.w if isinstance(w, str) else w)


def detach_featured_points(points: FeaturedPoints) -> FeaturedPoints:
    return FeaturedPoints(*(t.detach() if isinstance(t, torch.Tensor) else t for t in points))


def merge_featured_points(pcds: Union[List[FeaturedPoints], Tuple[FeaturedPoints, ...]]) -> FeaturedPoints:
    if not isinstance(pcds, (list, tuple)):
        raise ValueError()
    if any(p.w is not None for p in pcds):
        raise NotImplementedError          # gnn_data.py:55-57: weighted clouds cannot be merged
    return FeaturedPoints(x=torch.cat([p.x for p in pcds], dim=0), f=torch.cat([p.f for p in pcds], dim=0),
                          b=torch.cat([p.b for p in pcds], dim=0))


def flatten_featured_points(points: FeaturedPoints) -> FeaturedPoints:
    w = points.w.reshape(-1) if points.w is not None else None
    return FeaturedPoints(x=points.x.reshape(-1, 3), f=points.f.reshape(-1, points.f.shape[-1]),
                          b=points.b.reshape(-1), w=w)


def _cat_optional(a: Optional[torch.Tensor], b: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    if a is None or b is None:
        assert a is None and b is None
        return None
    return torch.cat([a, b], dim=0)


def cat_featured_points(fp1: FeaturedPoints, fp2: FeaturedPoints) -> FeaturedPoints:
    return FeaturedPoints(x=torch.cat([fp1.x, fp2.x], dim=0), f=torch.cat([fp1.f, fp2.f], dim=0), b=torch.cat([fp1.b, fp2.b], dim=0),
                          w=_cat_optional(fp1.w, fp2.w))


def set_graph_edge_attribute(graph_edge: GraphEdge, edge_src: Optional[torch.Tensor] = None, edge_dst: Optional[torch.Tensor] = None,
                             edge_length: Union[str, Optional[torch.Tensor]] = _KEEP, edge_attr: Union[str, Optional[torch.Tensor]] = _KEEP,
                             edge_scalars: Union[str, Optional[torch.Tensor]] = _KEEP, edge_weights: Union[str, Optional[torch.Tensor]] = _KEEP,
                             edge_logits: Union[str, Optional[torch.Tensor]] = _KEEP) -> GraphEdge:
    opt = dict(edge_length=edge_length, edge_attr=edge_attr, edge_scalars=edge_scalars, edge_weights=edge_weights, edge_logits=edge_logits)
    return GraphEdge(edge_src=graph_edge.edge_src if edge_src is None else edge_src,
                     edge_dst=graph_edge.edge_dst if edge_dst is None else edge_dst,
                     **{k: (getattr(graph_edge, k) if isinstance(v, str) else v) for k, v in opt.items()})


def cat_graph_edges(graph_edge_1: GraphEdge, graph_edge_2: GraphEdge) -> GraphEdge:
    return GraphEdge(edge_src=torch.cat([graph_edge_1.edge_src, graph_edge_2.edge_src], dim=0),
                     edge_dst=torch.cat([graph_edge_1.edge_dst, graph_edge_2.edge_dst], dim=0),
                     **{k: _cat_optional(getattr(graph_edge_1, k), getattr(graph_edge_2, k)) for k in GraphEdge._fields[2:]})
