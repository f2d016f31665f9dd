"""stats.py - statistics of sampled graphs from the bit matrix.

Counterpart of the degree / triangle / assortativity part of the reference's
`compute_graph_statistics` (eval_utils/graph_statistics.py:249-285), which `NetworkEvaluator`
(datasets/evaluator.py:27-37) applies to the 64 generated graphs of every evaluation.  The device
returns exact integer sums per graph; the floats are formed here in fp64 with the reference's formulas
(the reference routes triangles and assortativity through networkx, and its
`nx.from_scipy_sparse_matrix` no longer exists in networkx 3)."""
import ctypes as C

import numpy as np
import torch

from . import capi


STAT_KEYS = ('d_max', 'd_min', 'd', 'LCC', 'wedge_count', 'claw_count', 'triangle_count', 'square_count',
             'power_law_exp', 'gini', 'rel_edge_distr_entropy', 'assortativity', 'clustering_coefficient', 'cpl')


def power_law_alpha(degrees):
    """graph_statistics.py:146-161: `powerlaw.Fit(degrees, xmin=max(min(degrees), 1)).power_law.alpha`.  The `powerlaw`
    package is not in this image (version unpinned in the reference); this is its published estimator for that call
    (continuous data, fixed xmin; Clauset, Shalizi & Newman 2009, eq. 3.1): non-positive values are thrown out, the
    data below xmin cut, alpha = 1 + n / sum(ln(x / xmin)).  PARITY UNPINNED against the package itself."""
    d = np.asarray(degrees, dtype=np.float64)
    if d.size == 0:
        return np.nan
    xmin = max(float(d.min()), 1.0)
    d = d[d > 0]
    d = d[d >= xmin]
    with np.errstate(divide='ignore', invalid='ignore'):
        s = np.log(d / xmin).sum()
        return float(1.0 + d.size / s) if d.size else np.nan


def gini(degrees):
    """graph_statistics.py:164-181"""
    d = np.sort(np.asarray(degrees, dtype=np.float64))
    N = d.size
    with np.errstate(divide='ignore', invalid='ignore'):
        return float(2 * np.dot(d, np.arange(1, N + 1)) / (N * np.sum(d)) - (N + 1) / N) if N else np.nan


def edge_distribution_entropy(degrees):
    """graph_statistics.py:184-199 on a float degree vector (an isolated node gives log(0) * 0 = nan there too)."""
    d = np.asarray(degrees, dtype=np.float64)
    N = d.size
    with np.errstate(divide='ignore', invalid='ignore'):
        d = d / d.sum()
        return float(-np.dot(np.log(d), d) / np.log(N)) if N else np.nan


def graph_statistics(batch, adj=None, full=True):
    """dict of numpy arrays [B] with the 14 entries of the reference's `compute_graph_statistics`
    (eval_utils/graph_statistics.py:271-284) + num_edges, for `adj` (default: the batch's current state).
    Device: exact integer sums (degrees, triangles, degree products, common-neighbour pairs, BFS path sums, component
    sizes); host: the reference's float formulas in fp64, and the three degree-only statistics (power law, gini,
    entropy) from the device's degree vector.  full=False skips the O(n^2 words) ones (squares, path lengths)."""
    adj = batch.adj if adj is None else adj
    L, dev, B = batch.layout, batch.device, batch.num_graphs
    with batch._dev():
        s = capi.stream_ptr()
        deg = torch.empty(batch.num_nodes + 32, dtype=torch.int32, device=dev)
        out = torch.empty((B, capi.NSTAT), dtype=torch.int64, device=dev)
        capi.check(capi.lib.edgehip_degree_from_adj(C.byref(L), adj.data_ptr(), deg.data_ptr(), s), 'degree')
        capi.check(capi.lib.edgehip_graph_stats(C.byref(L), adj.data_ptr(), deg.data_ptr(), out.data_ptr(), s), 'graph_stats')
        ext = None
        if full:
            labels = torch.empty(batch.num_nodes + 32, dtype=torch.int32, device=dev)
            work = torch.empty(batch.num_nodes + 32, dtype=torch.int32, device=dev)
            ext = torch.empty((B, 4), dtype=torch.int64, device=dev)
            capi.check(capi.lib.edgehip_graph_stats_ext(C.byref(L), adj.data_ptr(), labels.data_ptr(), work.data_ptr(),
                                                        ext.data_ptr(), s), 'graph_stats_ext')
            batch.component_labels = labels[:batch.num_nodes]
    o = out.cpu().numpy()
    n = batch.nodes_per_graph.cpu().numpy().astype(np.float64)
    E, wedges, claws, tri3 = o[:, 0], o[:, 1], o[:, 2], o[:, 3]
    sdd, ssum, ssq = (o[:, k].astype(np.float64) for k in (4, 5, 6))
    M = 2.0 * E                                                    # directed edge ends
    with np.errstate(divide='ignore', invalid='ignore'):
        ex = ssum / M
        assort = (2.0 * sdd / M - ex * ex) / (ssq / M - ex * ex)   # Pearson r of the end degrees (networkx)
        clustering = tri3.astype(np.float64) / wedges              # 3 T / wedges   (graph_statistics.py:217-230)
        dmean = o[:, 9] / n
    res = dict(d_max=o[:, 7], d_min=np.where(n > 0, o[:, 8], 0), d=dmean, wedge_count=wedges, claw_count=claws,
               triangle_count=tri3 // 3, assortativity=assort, clustering_coefficient=clustering, num_edges=E)
    dh = deg[:batch.num_nodes].cpu().numpy()
    ptr = batch.host.node_ptr
    per = [dh[ptr[b]:ptr[b + 1]] for b in range(B)]
    res['power_law_exp'] = np.array([power_law_alpha(d) for d in per])
    res['gini'] = np.array([gini(d) for d in per])
    res['rel_edge_distr_entropy'] = np.array([edge_distribution_entropy(d) for d in per])
    if ext is not None:
        e = ext.cpu().numpy()
        with np.errstate(divide='ignore', invalid='ignore'):
            res['square_count'] = e[:, 0].astype(np.float64) / 4.0          # graph_statistics.py:141
            res['cpl'] = e[:, 1].astype(np.float64) / e[:, 2].astype(np.float64)
        res['LCC'] = e[:, 3]
    return res


class NetworkEvaluator:
    """datasets/evaluator.py:27-37: normalised mean absolute error of the 14 statistics of generated graphs against
    ONE reference network, plus their mean values.  `reference`: (n, edges[E,2]); `evaluate` takes a sampled
    GraphBatch (the state on the device; nothing goes through networkx)."""

    def __init__(self, reference, device='cuda'):
        from .batch import GraphBatch
        n, edges = reference
        rb = GraphBatch([n], edges=[np.asarray(edges)], device=device)
        self.reference_stats = {k: float(v[0]) for k, v in graph_statistics(rb).items() if k in STAT_KEYS}

    def evaluate(self, batch, adj=None):
        st = graph_statistics(batch, adj)
        metrics = {}
        with np.errstate(divide='ignore', invalid='ignore'):
            for k in STAT_KEYS:
                mean = float(np.mean(np.asarray(st[k], dtype=np.float64)))
                metrics[f'nmae/{k}'] = abs((mean - self.reference_stats[k]) / self.reference_stats[k])
                metrics[f'value/{k}'] = mean
        return metrics


def edge_overlap(batch, adj_a, adj_b):
    """Shared edges per graph of two states of the same batch (graph_statistics.py:287-299)."""
    out = torch.empty(batch.num_graphs, dtype=torch.int64, device=batch.device)
    capi.check(capi.lib.edgehip_edge_overlap(C.byref(batch.layout), adj_a.data_ptr(), adj_b.data_ptr(), out.data_ptr(),
                                             capi.stream_ptr()), 'edge_overlap')
    return (out // 2).cpu().numpy()
