"""Container helpers of the boundary (diffusion_edf_amd/gnn_data.py) against the behaviour of reference gnn_data.py:12-234."""
import pytest
import torch

from diffusion_edf_amd import gnn_data as G


def _fp(n, w=True, seed=0):
    g = torch.Generator().manual_seed(seed)
    return G.FeaturedPoints(torch.randn(n, 3, generator=g), torch.randn(n, 5, generator=g), torch.zeros(n, dtype=torch.long),
                            torch.rand(n, generator=g) if w else None)


def test_field_order_matches_the_reference():
    assert G.FeaturedPoints._fields == ('x', 'f', 'b', 'w')
    assert G.GraphEdge._fields == ('edge_src', 'edge_dst', 'edge_length', 'edge_attr', 'edge_scalars', 'edge_weights', 'edge_logits')


def test_set_attribute_keeps_or_replaces():
    p = _fp(4)
    q = G.set_featured_points_attribute(p, f=torch.ones(4, 5))
    assert q.x is p.x and q.b is p.b and q.w is p.w and bool((q.f == 1).all())
    assert G.set_featured_points_attribute(p, w=None).w is None            # None is a value; only the '' marker keeps the field
    e = G.GraphEdge(torch.arange(3), torch.arange(3), edge_logits=torch.zeros(3))
    e2 = G.set_graph_edge_attribute(e, edge_attr=torch.ones(3, 9), edge_logits=None)
    assert e2.edge_src is e.edge_src and e2.edge_attr.shape == (3, 9) and e2.edge_logits is None and e2.edge_length is None


def test_cat_merge_flatten_detach():
    a, b = _fp(3, seed=1), _fp(2, seed=2)
    c = G.cat_featured_points(a, b)
    assert c.x.shape == (5, 3) and torch.equal(c.w, torch.cat([a.w, b.w])) and torch.equal(c.f[3:], b.f)
    with pytest.raises(AssertionError):
        G.cat_featured_points(a, _fp(2, w=False))
    m = G.merge_featured_points([_fp(3, w=False), _fp(2, w=False)])
    assert m.x.shape == (5, 3) and m.w is None
    with pytest.raises(NotImplementedError):
        G.merge_featured_points((a, b))
    with pytest.raises(ValueError):
        G.merge_featured_points(iter([a]))
    p = G.FeaturedPoints(torch.zeros(2, 4, 3), torch.zeros(2, 4, 7), torch.zeros(2, 4, dtype=torch.long), torch.zeros(2, 4))
    fl = G.flatten_featured_points(p)
    assert fl.x.shape == (8, 3) and fl.f.shape == (8, 7) and fl.b.shape == (8,) and fl.w.shape == (8,)
    r = G.FeaturedPoints(torch.zeros(2, 3, requires_grad=True), torch.zeros(2, 5, requires_grad=True), torch.zeros(2, dtype=torch.long))
    d = G.detach_featured_points(r)
    assert not d.x.requires_grad and not d.f.requires_grad and d.w is None
    e1 = G.GraphEdge(torch.tensor([0, 1]), torch.tensor([2, 3]), edge_length=torch.ones(2))
    e2 = G.GraphEdge(torch.tensor([4]), torch.tensor([5]), edge_length=torch.zeros(1))
    e = G.cat_graph_edges(e1, e2)
    assert e.edge_src.tolist() == [0, 1, 4] and e.edge_length.tolist() == [1., 1., 0.] and e.edge_attr is None
    with pytest.raises(AssertionError):
        G.cat_graph_edges(e1, G.GraphEdge(torch.tensor([4]), torch.tensor([5])))
