"""Graph primitives of the feature extractors (dedf_fps / dedf_radius, diffusion_edf_amd/connectivity.py) against
oracle/graph_oracle.py.  Index work: the bar is bit-exact."""
import numpy as np
import pytest
import torch

from diffusion_edf_amd import synthetic
from oracle import graph_oracle as G


def _cloud(n, seed):
    return synthetic.make_scene(n, seed=seed).astype(np.float32) if hasattr(synthetic, "make_scene") else \
        np.random.default_rng(seed).uniform(-20, 20, size=(n, 3)).astype(np.float32)


# ---- oracle (CPU) ---------------------------------------------------------------------------------------------------------
def test_fps_oracle_properties():
    x = _cloud(500, 0)
    idx = G.fps(x, 0.2)
    assert len(idx) == 100 and idx[0] == 0 and len(set(idx.tolist())) == 100
    # each pick maximises the distance to the picks before it
    for i in (1, 2, 17, 99):
        d = np.min(((x[:, None, :] - x[idx[:i]][None, :, :]) ** 2).sum(-1), axis=1)
        assert d[idx[i]] == d.max()
    # the min-distance to the set never increases
    dmin = [np.min(((x[idx[i]] - x[idx[:i]]) ** 2).sum(-1)) for i in range(1, 100)]
    assert all(a >= b - 1e-9 for a, b in zip(dmin, dmin[1:]))
    # ratio rounding: ceil
    assert len(G.fps(x, 0.1001)) == 51 and len(G.fps(x[:7], 0.2)) == 2
    # same picks as the bench's own generator (float64 arithmetic) on a cloud without near-ties
    assert np.array_equal(idx, synthetic.fps(x.astype(np.float64), 0.2))
    # duplicates: once every distance is 0 the first index wins, like numpy argmax
    assert G.fps(np.zeros((4, 3), np.float32), 1.0).tolist() == [0, 0, 0, 0]


@pytest.mark.parametrize("case", ["random", "grid_ties", "duplicates"])
def test_fps_of_an_fps_ordered_cloud_is_its_prefix_on_the_oracle(case):
    """what lets a deterministic FPS cascade (UNet levels 1.., `FpsPool(_fps_ordered=True)`) skip its re-sampling: FPS of a cloud given in the
    selection order of an FPS that started at its first point returns 0, 1, 2, ... -- with exact distance ties (grid, duplicated points) too"""
    rng = np.random.default_rng(3)
    if case == "random":
        x = rng.normal(size=(3000, 3)).astype(np.float32)
    elif case == "grid_ties":
        g = np.arange(12, dtype=np.float32)
        x = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
        x = x[rng.permutation(len(x))]
    else:
        x = rng.normal(size=(800, 3)).astype(np.float32)
        x = np.concatenate([x, x[:300], x[100:200]], 0)
    for ratio in (0.2, 0.5):
        p0 = G.fps(x, ratio)
        sub = x[p0]
        for r1 in (0.2, 0.37, 1.0):
            p1 = G.fps(sub, r1)
            assert np.array_equal(p1, np.arange(len(p1))), (case, ratio, r1)


def test_fps_prefix_shortcut_ends_where_the_distinct_points_do():
    """the prefix property needs every prefix sample to have been at non-zero distance from the earlier ones.  A parent run that
    takes more samples than the cloud has distinct points selects index 0 again and again (all distances 0, smallest-index arg-max), and the
    re-sampling of THAT selection is no longer 0, 1, 2, ...: `FpsPool(_fps_ordered=True)` is documented as invalid there (DEDF_FPS_CHECK=1 raises)."""
    rng = np.random.default_rng(5)
    base = rng.normal(size=(50, 3)).astype(np.float32)
    x = np.concatenate([base, base, base], 0)                 # 150 points, 50 distinct
    p0 = G.fps(x, 0.5)                                        # 75 samples > 50 distinct points
    assert len(np.unique(x[p0], axis=0)) == 50 and len(p0) == 75
    sub = x[p0]
    p1 = G.fps(sub, 0.8)                                      # 60 samples of the sub-cloud: beyond its 50 distinct points
    assert np.array_equal(p1[:50], np.arange(50))             # the prefix holds exactly as far as the distinct samples go
    assert not np.array_equal(p1, np.arange(len(p1)))         # ... and not beyond


def test_radius_oracle_properties():
    x, y = _cloud(300, 1), _cloud(120, 2)
    ed, es = G.radius(x, y, 9.0, 1000)
    d = np.sqrt(((y[:, None, :].astype(np.float64) - x[None, :, :]) ** 2).sum(-1))
    far, near = d > 9.0 + 1e-4, d < 9.0 - 1e-4
    m = np.zeros_like(d, dtype=bool); m[ed, es] = True
    assert not (m & far).any() and m[near].all()
    assert np.all(np.diff(ed) >= 0) and np.all((np.diff(es) > 0) | (np.diff(ed) > 0))            # sorted by dst, then src
    ed3, es3 = G.radius(x, y, 9.0, 3)                                                           # cap: the first 3 sources of each dst
    for dd in range(len(y)):
        assert es3[ed3 == dd].tolist() == es[ed == dd][:3].tolist()
    gd, gs = G.radius(x, x, 6.0, 1000, exclude_self=True)                                        # radius_graph(loop=False): symmetric, no loops
    assert not (gd == gs).any()
    pairs = set(zip(gd.tolist(), gs.tolist()))
    assert all((b, a) in pairs for a, b in pairs)


# ---- HIP path (GPU) -------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("n,ratio", [(1, 1.0), (7, 0.2), (820, 0.2), (4096, 0.2), (5000, 0.03), (16384, 0.2), (40000, 0.01)])
def test_fps_bit_exact(n, ratio):
    from diffusion_edf_amd import connectivity as K
    x = _cloud(n, n)
    ref = G.fps(x, ratio)
    got = K.fps(torch.from_numpy(x).cuda(), None, ratio=ratio, random_start=False).cpu().numpy()
    assert got.dtype == np.int64 and np.array_equal(got, ref)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["grid_ties", "duplicates", "n1025", "n2049", "n4097", "n8192", "n12000_padding", "n16383", "line", "start_mid",
                                  "everything_n3000", "everything_n9000", "ten_distinct", "two_clusters", "grid_large"])
def test_fps_bucketed_kernel_bit_exact(case):
    """the bucketed + batched kernel (dedf_graph.h::k_fps_bucketed: Morton buckets skipped when the sample cannot lower any of their minima;
    from the 33rd sample on, batches of up to 64 samples drawn by one wave from the candidates above a threshold and applied to the buckets in
    one pass; what dedf_fps runs from 1 025 to 16 384 points: every register layout -- 16 / 32 points per thread, 4 / 8 waves -- is
    among the cases): identical to the exhaustive arg-max of the oracle.  Exact ties (grids, duplicated points: the batch's
    tie path and its fall-back to single samples when more points tie at the top than the candidate list holds), sampling EVERY point (the
    largest minimum reaches 0: no threshold below it), padding in the last bucket, degenerate extents, a start point in the middle."""
    from diffusion_edf_amd import connectivity as K
    from diffusion_edf_amd import _lib
    rng = np.random.default_rng(5)
    ratio, start = 0.2, 0
    if case == "grid_ties":
        x = np.stack(np.meshgrid(np.arange(16.), np.arange(16.), np.arange(8.), indexing="ij"), -1).reshape(-1, 3).astype(np.float32); ratio = 0.5
    elif case == "duplicates":
        base = _cloud(700, 3)
        x = base[rng.integers(0, 700, 5000)]; ratio = 0.3           # every point several times: ties at 0 and between copies
    elif case == "line":
        x = np.zeros((3000, 3), np.float32); x[:, 1] = rng.permutation(3000).astype(np.float32) * 0.25      # zero extent in two axes
    elif case == "start_mid":
        x = _cloud(6000, 9); start = 4321
    elif case.startswith("everything"):
        x = _cloud(int(case.split("_n")[1]), 21); ratio = 1.0
    elif case == "ten_distinct":
        x = _cloud(10, 4)[rng.integers(0, 10, 6000)]; ratio = 0.05    # 300 samples of 10 distinct points: from the 11th on every minimum is 0
    elif case == "two_clusters":
        x = np.concatenate([_cloud(7000, 1) * 0.01, _cloud(7000, 2) * 0.01 + 500.0]).astype(np.float32); ratio = 0.1
    elif case == "grid_large":
        x = np.stack(np.meshgrid(np.arange(32.), np.arange(32.), np.arange(16.), indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
        x = x[rng.permutation(len(x))]; ratio = 0.25                   # 16 384 lattice points: hundreds of exact ties at every level
    else:
        n = int(case[1:].split("_")[0])
        x = _cloud(n, n)
    ref = G.fps(x, ratio, start=start)
    if start == 0:
        got = K.fps(torch.from_numpy(x).cuda(), None, ratio=ratio, random_start=False).cpu().numpy()
    else:
        lib = _lib.load()
        xd = torch.from_numpy(x).cuda(); out = torch.empty(len(ref), dtype=torch.int32, device="cuda")
        assert lib.dedf_fps(xd.data_ptr(), len(x), len(ref), start, out.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
        got = out.cpu().numpy().astype(np.int64)
    assert np.array_equal(got, ref), (case, int(np.argmax(got != ref)))


@pytest.mark.gpu
@pytest.mark.parametrize("n", [3000, 8192, 16383])
def test_fps_exhaustive_kernel_bit_exact(n):
    """DEDF_FPS_BUCKETED=0 (read once per process: a subprocess): the exhaustive kernel at the sizes the batched one has taken over"""
    import os, subprocess, sys
    if os.environ.get("DEDF_FPS_BUCKETED") != "0":
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", f"{os.path.abspath(__file__)}::test_fps_exhaustive_kernel_bit_exact[{n}]"],
                           cwd=root, env=dict(os.environ, DEDF_FPS_BUCKETED="0"), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
        return
    from diffusion_edf_amd import connectivity as K
    x = _cloud(n, n)
    got = K.fps(torch.from_numpy(x).cuda(), None, ratio=0.2, random_start=False).cpu().numpy()
    assert np.array_equal(got, G.fps(x, 0.2))


@pytest.mark.gpu
def test_fps_of_an_fps_ordered_cloud_is_its_prefix():
    """the same on the kernels (plain and bucketed FPS), and FpsPool(_fps_ordered=True) against the sampled pool"""
    from diffusion_edf_amd import connectivity as CN
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(4)
    g = np.arange(10, dtype=np.float32)
    grid = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    for x in (rng.normal(size=(16384, 3)).astype(np.float32), grid[rng.permutation(len(grid))], synthetic.make_scene(9000, seed=2).astype(np.float32)):
        xt = torch.from_numpy(x).to(dev)
        p0 = CN.fps(xt, None, ratio=0.2, random_start=False)
        sub = xt[p0]
        p1 = CN.fps(sub, None, ratio=0.2, random_start=False)
        assert torch.equal(p1.cpu(), torch.arange(len(p1)))
        b = torch.zeros(len(sub), dtype=torch.long, device=dev)
        f = torch.randn(len(sub), 5, device=dev)
        pool = CN.FpsPool(ratio=0.2, random_start=False, r=2.5, max_num_neighbors=1000)
        a, c = pool(sub, f, b), pool(sub, f, b, _fps_ordered=True)
        assert all(torch.equal(u, v) for u, v in zip(a, c))
    # the cross-check switch: a cloud that is NOT in selection order is caught
    import os
    os.environ["DEDF_FPS_CHECK"] = "1"
    try:
        pool(sub, f, b, _fps_ordered=True)
        with pytest.raises(RuntimeError, match="not in the selection order"):
            pool(sub.flip(0).contiguous(), f, b, _fps_ordered=True)
    finally:
        del os.environ["DEDF_FPS_CHECK"]


@pytest.mark.gpu
def test_fps_ties_and_limits():
    from diffusion_edf_amd import connectivity as K
    grid = np.stack(np.meshgrid(np.arange(12.), np.arange(12.), np.arange(3.), indexing="ij"), -1).reshape(-1, 3).astype(np.float32)   # many exact ties
    assert np.array_equal(K.fps(torch.from_numpy(grid).cuda(), None, ratio=0.5, random_start=False).cpu().numpy(), G.fps(grid, 0.5))
    z = torch.zeros(5, 3, device="cuda")
    assert K.fps(z, None, ratio=1.0, random_start=False).tolist() == [0, 0, 0, 0, 0]
    assert K.fps(torch.zeros(9000, 3, device="cuda"), None, ratio=0.01, random_start=False).tolist() == [0] * 90      # (bucketed kernel: zero extent in every axis)
    with pytest.raises(NotImplementedError):
        K.fps(torch.zeros(70000, 3, device="cuda"), None, ratio=0.001, random_start=False)
    with pytest.raises(RuntimeError):
        K.fps(torch.zeros(5, 3), None, ratio=0.5, random_start=False)          # CPU tensor: no CPU path
    i = K.fps(torch.from_numpy(grid).cuda(), None, ratio=0.1, random_start=True)
    assert len(i) == 44 and len(set(i.tolist())) == 44


@pytest.mark.gpu
@pytest.mark.parametrize("ns,nd,r,cap", [(300, 120, 9.0, 1000), (300, 120, 9.0, 3), (4096, 820, 3.0, 1000), (5, 2000, 50.0, 1000), (1500, 1, 12.0, 7)])
def test_radius_bit_exact(ns, nd, r, cap):
    from diffusion_edf_amd import connectivity as K
    x, y = _cloud(ns, 3), _cloud(nd, 4)
    ed, es = G.radius(x, y, r, cap)
    e = K.radius(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda(), r, max_num_neighbors=cap).cpu().numpy()
    assert e.shape == (2, len(ed)) and np.array_equal(e[0], ed) and np.array_equal(e[1], es)


@pytest.mark.gpu
def test_connectivity_modules_match_oracle():
    """RadiusGraph / FpsPool as reference connectivity.py:8-80 composes them"""
    from diffusion_edf_amd import connectivity as K
    x = _cloud(3000, 5)
    xt = torch.from_numpy(x).cuda()
    f = torch.arange(len(x), dtype=torch.float32, device="cuda")[:, None].repeat(1, 4)
    b = torch.zeros(len(x), dtype=torch.long, device="cuda")
    # RadiusGraph
    fo, xo, es, ed, deg, bo = K.RadiusGraph(r=2.5, max_num_neighbors=1000)(xt, f, b)
    rd, rs = G.radius(x, x, 2.5, 1000, exclude_self=True)
    assert np.array_equal(ed.cpu().numpy(), rd) and np.array_equal(es.cpu().numpy(), rs)
    assert np.array_equal(deg.cpu().numpy(), np.bincount(rd, minlength=len(x))) and fo is f and xo is xt and bo is b
    # FpsPool: pooled nodes + bipartite edges without the pooled node itself
    fo, xo, es, ed, deg, bo = K.FpsPool(ratio=0.25, random_start=False, r=4.0, max_num_neighbors=1000)(xt, f, b)
    idx = G.fps(x, 0.25)
    rd, rs = G.radius(x, x[idx], 4.0, 1000)
    keep = idx[rd] != rs
    assert np.array_equal(xo.cpu().numpy(), x[idx]) and np.array_equal(fo[:, 0].cpu().numpy(), idx.astype(np.float32))
    assert np.array_equal(ed.cpu().numpy(), rd[keep]) and np.array_equal(es.cpu().numpy(), rs[keep])
    assert np.array_equal(deg.cpu().numpy(), np.bincount(rd[keep], minlength=len(idx))) and len(bo) == len(idx)


# ---- several clouds in one batch vector (reference connectivity.py:62 fps(batch=...), :43 radius(batch_x, batch_y)) -----------------------
def _three_clouds():
    xs = [_cloud(700, 11), _cloud(64, 12) + np.float32(3.0), _cloud(1500, 13)]
    x = np.concatenate(xs).astype(np.float32)
    batch = np.concatenate([np.full(len(c), b) for c, b in zip(xs, (0, 1, 4))])
    return xs, x, batch


def test_batched_oracle_is_the_per_cloud_oracle():
    xs, x, batch = _three_clouds()
    idx = G.fps_batched(x, batch, 0.1)
    off = np.cumsum([0] + [len(c) for c in xs])
    assert np.array_equal(idx, np.concatenate([G.fps(c, 0.1) + o for c, o in zip(xs, off)])) and len(idx) == 70 + 7 + 150
    ed, es = G.radius_batched(x, x[idx], 6.0, batch, batch[idx], 1000)
    assert np.all(batch[es] == batch[idx][ed]) and np.all(np.diff(ed) >= 0)                    # no pair crosses clouds; sorted by destination
    # the clouds overlap in space: without the batch vector there would be cross-cloud pairs
    ed1, es1 = G.radius(x, x[idx], 6.0, 1000)
    assert len(ed1) > len(ed)
    # a destination cloud whose id has no source cloud gets no edges
    ed2, _ = G.radius_batched(x[:700], x[idx], 6.0, batch[:700], batch[idx], 1000)
    assert set(np.unique(batch[idx][ed2]).tolist()) == {0}


@pytest.mark.gpu
def test_batched_graphs_bit_exact_and_unet_on_two_clouds():
    from diffusion_edf_amd import connectivity as K
    from diffusion_edf_amd.gnn_data import FeaturedPoints
    from diffusion_edf_amd.unet import UnetFeatureExtractor
    xs, x, batch = _three_clouds()
    tx, tb = torch.from_numpy(x).cuda(), torch.from_numpy(batch).cuda()
    idx = K.fps(tx, tb, ratio=0.1, random_start=False)
    assert np.array_equal(idx.cpu().numpy(), G.fps_batched(x, batch, 0.1))
    e = K.radius(tx, tx[idx], 6.0, tb, tb[idx], max_num_neighbors=1000)
    ed, es = G.radius_batched(x, x[idx.cpu().numpy()], 6.0, batch, batch[idx.cpu().numpy()], 1000)
    assert np.array_equal(e[0].cpu().numpy(), ed) and np.array_equal(e[1].cpu().numpy(), es)
    g = K.radius_graph(tx, 2.0, tb, loop=False, max_num_neighbors=1000)
    gd, gs = G.radius_batched(x, x, 2.0, batch, batch, 1000, exclude_self=True)
    assert np.array_equal(g[0].cpu().numpy(), gd) and np.array_equal(g[1].cpu().numpy(), gs)
    with pytest.raises(ValueError, match="sorted"):
        K.fps(tx, torch.flip(tb, dims=[0]), ratio=0.1, random_start=False)
    # the whole extractor on two clouds in one batch vector = the two clouds one after the other
    m = UnetFeatureExtractor(**synthetic.unet_kwargs("panda_lowres"), deterministic=True).cuda()
    a, b = torch.from_numpy(_cloud(1200, 21)).cuda(), torch.from_numpy(_cloud(800, 22)).cuda()
    fa, fb = torch.rand(len(a), 3, device="cuda"), torch.rand(len(b), 3, device="cuda")
    z = lambda n, v: torch.full((n,), v, dtype=torch.long, device="cuda")
    both = m(FeaturedPoints(x=torch.cat([a, b]), f=torch.cat([fa, fb]), b=torch.cat([z(len(a), 0), z(len(b), 1)]), w=None))
    oa, ob = m(FeaturedPoints(x=a, f=fa, b=z(len(a), 0), w=None)), m(FeaturedPoints(x=b, f=fb, b=z(len(b), 0), w=None))
    for lv, (pa, pb) in enumerate(zip(oa, ob)):
        na = len(pa.x)
        assert torch.equal(both[lv].x[:na], pa.x) and torch.equal(both[lv].x[na:], pb.x) and int(both[lv].b[:na].sum()) == 0 and bool((both[lv].b[na:] == 1).all())
        ref = torch.cat([pa.f, pb.f])
        assert float((both[lv].f - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
