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


def graph_statistics(batch, adj=None):
    """dict of numpy arrays [B]: d_max, d_min, d (mean degree), wedge_count, claw_count, triangle_count,
    assortativity, clustering_coefficient, num_edges - for `adj` (default: the batch's current state)."""
    adj = batch.adj if adj is None else adj
    L, dev, B = batch.layout, batch.device, batch.num_graphs
    s = capi.stream_ptr()
    deg = torch.empty(batch.num_nodes + 32, dtype=torch.int32, device=dev)
    out = torch.empty((B, capi.NSTAT), dtype=torch.int64, device=dev)
    capi.check(capi.lib.edgehip_degree_from_adj(C.byref(L), adj.data_ptr(), deg.data_ptr(), s), 'degree')
    capi.check(capi.lib.edgehip_graph_stats(C.byref(L), adj.data_ptr(), deg.data_ptr(), out.data_ptr(), s), 'graph_stats')
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
    return dict(d_max=o[:, 7], d_min=np.where(n > 0, o[:, 8], 0), d=dmean, wedge_count=wedges, claw_count=claws,
                triangle_count=tri3 // 3, assortativity=assort, clustering_coefficient=clustering, num_edges=E)


def edge_overlap(batch, adj_a, adj_b):
    """Shared edges per graph of two states of the same batch (graph_statistics.py:287-299)."""
    out = torch.empty(batch.num_graphs, dtype=torch.int64, device=batch.device)
    capi.check(capi.lib.edgehip_edge_overlap(C.byref(batch.layout), adj_a.data_ptr(), adj_b.data_ptr(), out.data_ptr(),
                                             capi.stream_ptr()), 'edge_overlap')
    return (out // 2).cpu().numpy()
