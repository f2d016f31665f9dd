"""Synthetic input graphs shaped like the reference's datasets.

The reference ships its networks as networkx pickles (`graphs/*.pkl`) and lists
`Ego.pkl` / `community.pkl` as missing blobs.  Pickles are never loaded by this
project, so every workload here is a fixed-seed synthetic graph with the
published shape of the dataset it stands in for:

  polblogs      n 1222, E 16714, d_max ~351   (datasets/data.py:63-75 'polblogs')
  cora          n 2485, E 5069,  d_max ~168
  Homo_sapiens  n 3852, E 37841, d_max ~593
  community     2-block graphs, n in [12, 20]  (datasets/data.py:77-119 'community')
  Ego           ego-nets, n in [50, 400]
  SBM           20 blocks x 1000 nodes, p_in 0.02, p_out 5e-4 (BASELINE.json configs[4])

All generators are pure numpy (PCG64), so the same seed gives the same graph on
every machine.  Edges are returned as an int64 array [E,2] with i<j, sorted in
the reference's pair order (row-major over the strict upper triangle).
"""
import numpy as np


def _finish(n, pairs):
    pairs = np.asarray(pairs, dtype=np.int64).reshape(-1, 2)
    lo = np.minimum(pairs[:, 0], pairs[:, 1])
    hi = np.maximum(pairs[:, 0], pairs[:, 1])
    keep = lo != hi
    key = np.unique(lo[keep] * n + hi[keep])
    return np.stack([key // n, key % n], axis=1)


def degrees(n, edges):
    d = np.zeros(n, dtype=np.int64)
    if len(edges):
        np.add.at(d, edges[:, 0], 1)
        np.add.at(d, edges[:, 1], 1)
    return d


def power_law_graph(n, num_edges, d_max, seed):
    """Chung-Lu style graph with exactly `num_edges` edges and realised max degree ~d_max.

    Duplicate rejection flattens the hubs, so the expected-degree target is
    re-tuned a few times until the realised maximum is within 3% of d_max."""
    target = float(d_max)
    for _ in range(6):
        n_, e = _power_law_graph(n, num_edges, target, seed)
        got = degrees(n, e).max()
        if abs(got - d_max) <= 0.03 * d_max:
            break
        target *= d_max / got
    return n_, e


def _power_law_graph(n, num_edges, d_max, seed):
    rng = np.random.default_rng(seed)
    # weights w_i ~ (i + i0)^-gamma with sum 2E and max d_max: solve i0 for gamma by bisection
    gamma = 0.9
    two_e = 2.0 * num_edges

    def wts(i0):
        w = (np.arange(n) + i0) ** (-gamma)
        return w * (two_e / w.sum())
    lo, hi = 1e-3, 1e4
    for _ in range(200):
        mid = np.sqrt(lo * hi)
        if wts(mid)[0] > d_max:
            lo = mid
        else:
            hi = mid
    w = wts(hi)
    p = w / w.sum()
    have = np.zeros(0, dtype=np.int64)
    while len(have) < num_edges:
        m = int((num_edges - len(have)) * 1.3) + 64
        a = rng.choice(n, size=m, p=p)
        b = rng.choice(n, size=m, p=p)
        lo_, hi_ = np.minimum(a, b), np.maximum(a, b)
        k = (lo_ * n + hi_)[lo_ != hi_]
        # keep first occurrences in draw order, then trim
        allk = np.concatenate([have, k])
        _, first = np.unique(allk, return_index=True)
        have = allk[np.sort(first)][:num_edges]
    perm = rng.permutation(n)          # node ids carry no degree order
    e = np.stack([perm[have // n], perm[have % n]], axis=1)
    return n, _finish(n, e)


def polblogs_like(seed=0):
    return power_law_graph(1222, 16714, 351, seed)


def cora_like(seed=0):
    return power_law_graph(2485, 5069, 168, seed)


def homo_sapiens_like(seed=0):
    return power_law_graph(3852, 37841, 593, seed)


def sbm_graph(blocks, block_size, p_in, p_out, seed):
    rng = np.random.default_rng(seed)
    n = blocks * block_size
    out = []
    for bi in range(blocks):
        for bj in range(bi, blocks):
            p = p_in if bi == bj else p_out
            total = block_size * block_size
            m = rng.binomial(total, p)
            idx = rng.choice(total, size=m, replace=False)
            i = idx // block_size + bi * block_size
            j = idx % block_size + bj * block_size
            if bi == bj:
                keep = i < j
                i, j = i[keep], j[keep]
            out.append(np.stack([i, j], axis=1))
    return n, _finish(n, np.concatenate(out))


def community_like(num_graphs=100, seed=0):
    """2-community graphs with n in [12, 20] (stand-in for graphs/community.pkl)."""
    rng = np.random.default_rng(seed)
    gs = []
    for _ in range(num_graphs):
        n = int(rng.integers(12, 21))
        half = n // 2
        blk = (np.arange(n) >= half).astype(np.int64)
        iu, ju = np.triu_indices(n, 1)
        p = np.where(blk[iu] == blk[ju], 0.7, 0.05)
        keep = rng.random(len(iu)) < p
        e = np.stack([iu[keep], ju[keep]], axis=1)
        # one bridge so the graph is connected-ish
        e = np.concatenate([e, np.array([[0, n - 1]])])
        gs.append((n, _finish(n, e)))
    return gs


def ego_like(num_graphs=16, seed=0, n_lo=50, n_hi=400):
    """Ego-net shaped graphs (stand-in for graphs/Ego.pkl): a hub plus a sparse power-law rest."""
    rng = np.random.default_rng(seed)
    gs = []
    for g in range(num_graphs):
        n = int(rng.integers(n_lo, n_hi + 1))
        m = int(n * rng.uniform(1.5, 3.0))
        _, e = power_law_graph(n, m, max(8, n // 6), int(rng.integers(1 << 30)))
        hub = np.stack([np.zeros(n - 1, dtype=np.int64), np.arange(1, n)], axis=1)
        sel = rng.random(n - 1) < 0.5
        gs.append((n, _finish(n, np.concatenate([e, hub[sel]]))))
    return gs
