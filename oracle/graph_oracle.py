"""TEST INFRASTRUCTURE ONLY (imported by tests/, never by the product).

CPU restatement of the two torch_cluster primitives reference diffusion_edf/connectivity.py builds its graphs from
(torch_cluster is an un-vendored, unpinned dependency — setup.py lists it without a version — and not installable here, so
these semantics are **parity unpinned**: they restate the published algorithm as recalled and are anchored on the reference's
call sites):

  fps(src, batch, ratio, random_start)          connectivity.py:62   farthest point sampling: ceil(ratio*N) points, the first one
                                                                   is point 0 when random_start=False, each next one maximises
                                                                   the distance to the set chosen so far (first index on ties)
  radius(x, y, r, batch_x, batch_y, max_num_neighbors)  :43        for every y all x with |x - y| < r, at most max_num_neighbors
                                                                   (the first ones in x order); returns (y index, x index)
  radius_graph(x, r, batch, loop=False, max_num_neighbors) :22     radius(x, x) without the self pairs

All arithmetic is float32 with one rounding per operation, d2 = ((dx*dx + dy*dy) + dz*dz), so that an implementation using the
same sequence reproduces every index bit for bit."""
import math

import numpy as np


def _d2(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    a = a.astype(np.float32, copy=False)
    b = b.astype(np.float32, copy=False)
    dx, dy, dz = a[..., 0] - b[..., 0], a[..., 1] - b[..., 1], a[..., 2] - b[..., 2]
    return (dx * dx + dy * dy) + dz * dz          # float32 arrays: every operation rounds to fp32, no contraction


def fps(x: np.ndarray, ratio: float, start: int = 0) -> np.ndarray:
    n = x.shape[0]
    k = int(math.ceil(ratio * n))
    idx = np.zeros(k, dtype=np.int64)
    md = np.full(n, np.inf, dtype=np.float32)
    cur = start
    for i in range(k):
        idx[i] = cur
        md = np.minimum(md, _d2(x, x[cur][None, :]))
        cur = int(np.argmax(md))                  # first maximum
    return idx


def radius(x_src: np.ndarray, x_dst: np.ndarray, r: float, max_num_neighbors: int, exclude_self: bool = False):
    """-> (edge_dst, edge_src) int64, sorted by dst then src"""
    r2 = np.float32(np.float32(r) * np.float32(r))
    ed, es = [], []
    for s in range(0, x_dst.shape[0], 2048):
        y = x_dst[s:s + 2048]
        m = _d2(y[:, None, :], x_src[None, :, :]) < r2
        if exclude_self:
            rows = np.arange(s, s + len(y))
            ok = rows < x_src.shape[0]
            m[np.nonzero(ok)[0], rows[ok]] = False
        m &= np.cumsum(m, axis=1) <= max_num_neighbors
        di, si = np.nonzero(m)
        ed.append(di + s)
        es.append(si)
    return np.concatenate(ed).astype(np.int64), np.concatenate(es).astype(np.int64)


# ---- several clouds in one (sorted) batch vector: torch_cluster builds its graphs cloud by cloud ----------------------------------
def _segments(batch: np.ndarray):
    batch = np.asarray(batch)
    assert np.all(np.diff(batch) >= 0), "batch vector must be sorted"
    ids, starts = np.unique(batch, return_index=True)
    ends = list(starts[1:]) + [len(batch)]
    return [(int(b), int(a), int(e)) for b, a, e in zip(ids, starts, ends)]


def fps_batched(x: np.ndarray, batch: np.ndarray, ratio: float) -> np.ndarray:
    """fps(src, batch, ratio, random_start=False): ceil(ratio * n_b) points of every cloud b, first pick = the cloud's first point"""
    return np.concatenate([fps(x[a:e], ratio) + a for _, a, e in _segments(batch)])


def radius_batched(x_src, x_dst, r, batch_src, batch_dst, max_num_neighbors, exclude_self: bool = False):
    """radius(x, y, r, batch_x, batch_y, ...): pairs exist inside one cloud only -> (edge_dst, edge_src), sorted by dst then src"""
    seg_s = {b: (a, e) for b, a, e in _segments(batch_src)}
    ed, es = [np.zeros(0, np.int64)], [np.zeros(0, np.int64)]
    for b, a, e in _segments(batch_dst):
        if b not in seg_s:
            continue
        sa, se = seg_s[b]
        d, s_ = radius(x_src[sa:se], x_dst[a:e], r, max_num_neighbors, exclude_self)
        ed.append(d + a); es.append(s_ + sa)
    return np.concatenate(ed), np.concatenate(es)
