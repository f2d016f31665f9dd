"""Empty-graph sampler: the step before the hot path (datasets/data_utils.py:78-147 of the
reference, `EmpiricalEmptyGraphGenerator`).  Draws a graph size from the empirical size
distribution of the training graphs and a target degree sequence d0 from a random training
graph of that size, and emits the native GraphBatch (node_ptr / pair_ptr / d0) directly -
no O(n^2) `triu_indices`, no `Batch.from_data_list`.
"""
import random

import numpy as np
import torch

from . import synth
from .batch import GraphBatch


class EmpiricalEmptyGraphGenerator:
    def __init__(self, train_graphs, degree=True, device='cuda', gid_base=0):
        """train_graphs: list of (n, edges[E,2]) with local ids."""
        sizes = torch.tensor([g[0] for g in train_graphs])
        self.min_node = int(sizes.min())
        self.max_node = int(sizes.max())
        unnorm_p = torch.histc(sizes.float(), bins=self.max_node - self.min_node + 1)
        self.empirical_graph_size_dist = unnorm_p / unnorm_p.sum()
        self.degree = degree
        self.device = device
        self.gid_base = gid_base
        self.empirical_node_feat_dist = {}
        for n, e in train_graphs:
            self.empirical_node_feat_dist.setdefault(n, []).append(
                {'degree': synth.degrees(n, np.asarray(e)).astype(np.float32)})

    def _sample_graph_size_and_features(self, num_samples):
        ret = self.empirical_graph_size_dist.multinomial(num_samples=num_samples, replacement=True) + self.min_node
        ret = ret.tolist()
        feats = [random.choice(self.empirical_node_feat_dist[n]) for n in ret]
        return ret, feats

    def sample(self, num_samples):
        sizes, feats = self._sample_graph_size_and_features(num_samples)
        return GraphBatch(sizes, degree=[f['degree'] for f in feats], device=self.device, gid_base=self.gid_base)
