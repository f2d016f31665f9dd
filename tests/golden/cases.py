"""Seeded inputs shared by the golden-vector generator and the parity tests.

Only inputs that can be regenerated from a seed live here (graphs, weights,
uniforms); the expected OUTPUTS are what `make_golden.py` stores as `.npz`.
Uniforms are drawn on the 2^-24 grid that `torch.rand` (CPU, fp32) emits.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'graph-generation-edge_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

from edgehip import synth  # noqa: E402

GOLDEN_DIR = os.path.dirname(os.path.abspath(__file__))
HEADS = (8, 8, 8, 8, 1)
DIM = 64


def uniforms(rng, *shape):
    return torch.from_numpy((rng.integers(0, 1 << 24, size=shape).astype(np.float32) * np.float32(2.0 ** -24)))


def random_graph(rng, n, p):
    iu, ju = np.triu_indices(n, 1)
    keep = rng.random(len(iu)) < p
    return np.stack([iu[keep], ju[keep]], axis=1).astype(np.int64)


def toy_graphs(seed=11):
    """Two small graphs (n=7, n=9) - the 2-graph toy batch of SURVEY §8c."""
    rng = np.random.default_rng(seed)
    return [dict(n=7, edges=random_graph(rng, 7, 0.5)), dict(n=9, edges=random_graph(rng, 9, 0.45))]


def ragged_graphs(seed=12):
    """Ragged batch incl. a 1-node graph (no pairs), a 2-node graph and an empty graph."""
    rng = np.random.default_rng(seed)
    return [dict(n=1, edges=np.zeros((0, 2), np.int64)),
            dict(n=12, edges=random_graph(rng, 12, 0.4)),
            dict(n=2, edges=np.array([[0, 1]], np.int64)),
            dict(n=33, edges=random_graph(rng, 33, 0.2)),
            dict(n=5, edges=np.zeros((0, 2), np.int64)),
            dict(n=70, edges=random_graph(rng, 70, 0.15))]


def polblogs_graph(seed=0):
    n, e = synth.polblogs_like(seed)
    return [dict(n=n, edges=e)]


def max_degree_of(graphs):
    return int(max(synth.degrees(g['n'], g['edges']).max() if len(g['edges']) else 0 for g in graphs))


CASES = {
    # name: (graph builder, T, weight seed, uniform seed)
    'toy': (toy_graphs, 32, 101, 201),
    'ragged': (ragged_graphs, 48, 102, 202),
    'polblogs1': (polblogs_graph, 512, 103, 203),
}


def dropout_masks(useed, num_nodes, p, layers=len(HEADS), dim=DIM):
    """Keep-masks of the training-mode dropout, one [Nn, dim] 0/1 array per call of `self.dropout` in
    TGNN_degree_guided.forward (layers.py:357,361): slot 2i = conv output of layer i, slot 2i+1 = its GRU hidden."""
    rng = np.random.default_rng(useed + 7)
    return [torch.from_numpy((rng.random((num_nodes, dim)) >= p).astype(np.uint8)) for _ in range(2 * layers)]


def stats_graphs():
    """Graphs of the statistics fixture (stats.npz): ragged incl. degenerate ones, a sparse random graph with several
    components, a path (long diameter), two disjoint cliques, and the polblogs shape."""
    rng = np.random.default_rng(41)
    gs = [g for g in ragged_graphs(6)]
    gs.append(dict(n=150, edges=random_graph(rng, 150, 0.012)))
    gs.append(dict(n=40, edges=np.stack([np.arange(39), np.arange(1, 40)], 1).astype(np.int64)))
    a = np.stack(np.triu_indices(9, 1), 1)
    gs.append(dict(n=21, edges=np.concatenate([a, a + 9]).astype(np.int64)))          # 2 cliques + 3 isolated nodes
    gs.append(polblogs_graph(0)[0])
    return gs


def sampler_train_graphs():
    """Training set of the empty-graph sampler fixture: repeated sizes (so `random.choice` matters), a gap in the
    size histogram, one graph without edges."""
    rng = np.random.default_rng(51)
    sizes = [12, 33, 12, 70, 33, 12, 5, 33, 40]
    ps = [0.4, 0.2, 0.3, 0.15, 0.25, 0.5, 0.0, 0.1, 0.2]
    return [dict(n=n, edges=random_graph(rng, n, p) if p > 0 else np.zeros((0, 2), np.int64)) for n, p in zip(sizes, ps)]


SAMPLER_SEED, SAMPLER_DRAWS = 123, 25
