"""GraphBatch: the native counterpart of the PyG `Batch` objects the reference path consumes.

It carries the fields SURVEY.md section 8b lists (`degree`, `batch`, `nodes_per_graph`,
`edges_per_graph`, `num_nodes`, `num_graphs`, and after sampling `edge_index`, `edge_attr`,
`node_attr`, `_slice_dict['edge_index']`), but never materialises the O(n^2)
`full_edge_index` / `full_edge_attr` (datasets/data_utils.py:53-56): the edge state lives in a
symmetric bit matrix on the device (include/edgehip.h "State layout").
`full_edge_index` / `full_edge_attr` are provided as lazy properties for small batches so that
code written against the reference layout (tests, the oracle) can still read them.
"""
from types import SimpleNamespace

import numpy as np
import torch

from . import capi


class GraphBatch:
    def __init__(self, nodes_per_graph, degree=None, edges=None, device='cuda', gid_base=0, degree_dtype=None):
        """nodes_per_graph: list[int]; degree: list of per-graph degree arrays (d0) or None;
        edges: list of per-graph [E,2] local-id arrays (x0, training batches) or None."""
        npg = np.asarray(nodes_per_graph, dtype=np.int64)
        B = len(npg)
        self.device = torch.device(device)
        if self.device.type == 'cuda' and self.device.index is None:
            # 'cuda' and 'cuda:0' compare unequal: keep the resolved form, the one tensors report
            self.device = torch.device('cuda', torch.cuda.current_device())
        self.num_graphs = B
        self.num_nodes = int(npg.sum())
        node_ptr = np.zeros(B + 1, dtype=np.int64)
        node_ptr[1:] = np.cumsum(npg)
        stride = ((npg + 31) // 32 + 3) // 4 * 4
        stride = np.maximum(stride, 4)
        adj_ptr = np.zeros(B + 1, dtype=np.int64)
        adj_ptr[1:] = np.cumsum(npg * stride)
        pair_ptr = np.zeros(B + 1, dtype=np.int64)
        pair_ptr[1:] = np.cumsum(npg * (npg - 1) // 2)
        tiles = (npg + 31) // 32
        tile_ptr = np.zeros(B + 1, dtype=np.int64)
        tile_ptr[1:] = np.cumsum(tiles)
        NT = int(tile_ptr[-1])
        tile_graph = np.repeat(np.arange(B), tiles)
        tile_local = np.arange(NT) - tile_ptr[tile_graph] if NT else np.zeros(0, np.int64)
        tile_row0 = node_ptr[tile_graph] + tile_local * 32 if NT else np.zeros(0, np.int64)
        tile_rows = np.minimum(32, npg[tile_graph] - tile_local * 32) if NT else np.zeros(0, np.int64)
        node_graph = np.repeat(np.arange(B), npg)
        self.host = SimpleNamespace(nodes_per_graph=npg, node_ptr=node_ptr, adj_stride=stride, adj_ptr=adj_ptr,
                                    pair_ptr=pair_ptr, tile_ptr=tile_ptr)
        self.adj_words = int(adj_ptr[-1])
        self.num_tiles = NT
        self.num_pairs = int(pair_ptr[-1])
        dev = self.device

        def i32(a):
            return torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(dev)

        def i64(a):
            return torch.from_numpy(np.ascontiguousarray(a, dtype=np.int64)).to(dev)
        self._t = dict(node_ptr=i32(node_ptr), node_graph=i32(node_graph), adj_ptr=i64(adj_ptr),
                       adj_stride=i32(stride), pair_ptr=i64(pair_ptr), tile_ptr=i32(tile_ptr),
                       tile_graph=i32(tile_graph), tile_row0=i32(tile_row0), tile_rows=i32(tile_rows))
        L = capi.Layout()
        L.num_graphs, L.num_nodes, L.num_tiles = B, self.num_nodes, NT
        L.max_nodes = int(npg.max()) if B else 0
        L.gid_base = gid_base
        for k, v in self._t.items():
            setattr(L, k, v.data_ptr())
        self.layout = L
        # reference-named fields
        self.nodes_per_graph = torch.from_numpy(npg).to(dev)
        self.edges_per_graph = torch.from_numpy(npg * (npg - 1) // 2).to(dev)
        self.batch = self._t['node_graph'].long()
        self.node_attr = torch.zeros(self.num_nodes, dtype=torch.long, device=dev)
        if degree is not None:
            d = np.concatenate([np.asarray(x).reshape(-1) for x in degree]) if B else np.zeros(0)
            assert d.shape[0] == self.num_nodes
            self.degree_i32 = i32(np.rint(d))
            # dtype as in the reference: fp32 from the empty-graph sampler (data_utils.py:104),
            # int64 from preprocess (data_utils.py:62)
            self.degree = torch.from_numpy(d.astype(np.float32 if degree_dtype is None else degree_dtype)).to(dev)
        else:
            self.degree_i32 = None
            self.degree = None
        # state
        self.adj = torch.zeros(max(self.adj_words, 1), dtype=torch.int32, device=dev)
        self.x0_edges = edges
        if edges is not None:
            ge = [np.asarray(e, dtype=np.int64).reshape(-1, 2) + node_ptr[b] for b, e in enumerate(edges)]
            ge = np.concatenate(ge) if ge else np.zeros((0, 2), np.int64)
            self.set_edges(self.adj, ge)
            if degree is None:
                deg = torch.empty(self.num_nodes, dtype=torch.int32, device=dev)
                with self._dev():
                    capi.check(capi.lib.edgehip_degree_from_adj(C_byref(L), self.adj.data_ptr(), deg.data_ptr(),
                                                                capi.stream_ptr()), 'degree_from_adj')
                self.degree_i32 = deg
                self.degree = deg.long()
        # PyG bookkeeping the reference's epilogue fixes up (diffusion_base.py:227-228): per-graph offsets that
        # `separate` subtracts from an index attribute = the first node of each graph
        self._slice_dict, self._inc_dict = {}, {'full_edge_index': self._t['node_ptr'][:-1].long()}

    # ---- helpers -------------------------------------------------------------------------
    def _dev(self):
        """Context with this batch's GPU current (the library launches on the current device's current stream)."""
        import contextlib
        return torch.cuda.device(self.device) if self.device.type == 'cuda' else contextlib.nullcontext()

    def set_edges(self, adj, global_edges):
        with self._dev():
            return self._set_edges(adj, global_edges)

    def _set_edges(self, adj, global_edges):
        ge = torch.from_numpy(np.ascontiguousarray(global_edges, dtype=np.int32)).to(self.device)
        capi.check(capi.lib.edgehip_adj_from_edges(C_byref(self.layout), capi.ptr(ge) if ge.numel() else None,
                                                   ge.shape[0], adj.data_ptr(), capi.stream_ptr()), 'adj_from_edges')

    def new_adj(self):
        return torch.zeros_like(self.adj)

    def split(self, parts):
        """`parts` sub-batches over contiguous blocks of graphs that SHARE this batch's edge state (their `adj` are
        views of `self.adj`; d0 is copied, it is constant).  Graph g of block k keeps its global id, so the
        counter-based uniforms are those of the whole batch.  Used by the sampler to run blocks of graphs on
        separate HIP streams; cached."""
        cache = self.__dict__.setdefault('_split_cache', {})
        # the blocks alias self.adj and copy d0: a caller that rebinds either gets fresh blocks
        stamp = (self.adj.data_ptr(), None if self.degree_i32 is None else
                 (self.degree_i32.data_ptr(), self.degree_i32._version))
        if cache.get('stamp') != stamp:
            cache.clear()
            cache['stamp'] = stamp
        if parts in cache:
            return cache[parts]
        B = self.num_graphs
        parts = max(1, min(parts, B))
        sizes = [B // parts + (1 if i < B % parts else 0) for i in range(parts)]
        subs, g0 = [], 0
        npg = self.host.nodes_per_graph
        for k in sizes:
            g1 = g0 + k
            sub = GraphBatch(npg[g0:g1], degree=None, device=self.device, gid_base=int(self.layout.gid_base) + g0)
            w0, w1 = int(self.host.adj_ptr[g0]), int(self.host.adj_ptr[g1])
            n0, n1 = int(self.host.node_ptr[g0]), int(self.host.node_ptr[g1])
            if w1 > w0:
                sub.adj = self.adj[w0:w1]
            if self.degree_i32 is not None:
                sub.degree_i32 = self.degree_i32[n0:n1].clone()
                sub.degree = self.degree[n0:n1].clone()
            subs.append(sub)
            g0 = g1
        cache[parts] = subs
        return subs

    def to(self, device):
        assert torch.device(device).type == self.device.type, 'GraphBatch is built on its device'
        return self

    def extract_edges(self, adj=None):
        """edge_index [2,E] int64 (i<j, ascending pair order) and edge_slice [B+1] of a state."""
        with self._dev():
            return self._extract_edges(self.adj if adj is None else adj)

    def _extract_edges(self, adj):
        row_upper = torch.empty(max(self.num_nodes, 1), dtype=torch.int32, device=self.device)
        edge_slice = torch.zeros(self.num_graphs + 1, dtype=torch.int64, device=self.device)
        s = capi.stream_ptr()
        capi.check(capi.lib.edgehip_count_edges(C_byref(self.layout), adj.data_ptr(), row_upper.data_ptr(),
                                                edge_slice.data_ptr(), s), 'count_edges')
        E = int(edge_slice[-1].item())
        edge_index = torch.empty((2, E), dtype=torch.int64, device=self.device)
        capi.check(capi.lib.edgehip_extract_edges(C_byref(self.layout), adj.data_ptr(), row_upper.data_ptr(),
                                                  edge_slice.data_ptr(), capi.ptr(edge_index) if E else None, E, s),
                   'extract_edges')
        return edge_index, edge_slice

    # ---- reference-layout views (small batches only) --------------------------------------
    @property
    def full_edge_index(self):
        rows, cols = [], []
        for b, n in enumerate(self.host.nodes_per_graph.tolist()):
            r, c = torch.triu_indices(n, n, 1)
            rows.append(r + int(self.host.node_ptr[b]))
            cols.append(c + int(self.host.node_ptr[b]))
        return torch.stack([torch.cat(rows), torch.cat(cols)]).to(self.device)

    def pair_bits(self, adj=None):
        """Dense int64 [P] bit vector of a state in the reference's pair order (tests)."""
        ei, _ = self.extract_edges(adj)
        out = torch.zeros(self.num_pairs, dtype=torch.long, device=self.device)
        if ei.shape[1]:
            g = self.batch[ei[0]]
            node0 = self._t['node_ptr'].long()[g]
            n = self.nodes_per_graph[g]
            i, j = ei[0] - node0, ei[1] - node0
            k = self._t['pair_ptr'][g] + i * (2 * n - i - 1) // 2 + (j - i - 1)
            out[k] = 1
        return out

    @property
    def full_edge_attr(self):
        return self.pair_bits()

    def local_edge_index(self):
        """edge_index with LOCAL node ids ([2,E] int64 on the device, one pass): what PyG's `separate` produces per
        graph by subtracting `_inc_dict['edge_index']` (diffusion_base.py:221-228)."""
        sl = self._slice_dict['edge_index']
        inc = self._inc_dict['edge_index']
        per_edge = torch.repeat_interleave(inc, sl[1:] - sl[:-1], output_size=int(self.edge_index.shape[1]))
        return self.edge_index - per_edge[None, :]

    def to_data_list(self, device=None):
        """Per-graph results with local node ids, like PyG's `Batch.to_data_list()` (consumer: experiment.py:70,
        evaluate.py:51).  The per-graph `edge_index` / `edge_attr` / `node_attr` are VIEWS of three tensors
        (no per-graph copies): on the batch's device with device='cuda' / self.device, else (default) of ONE pinned
        host copy of the whole edge list - a single D2H transfer however many graphs there are."""
        sl = self._slice_dict['edge_index'].cpu().tolist()
        ei = self.local_edge_index()
        ea, na = self.edge_attr, self.node_attr
        if device is None or torch.device(device).type == 'cpu':
            if ei.is_cuda:
                host = torch.empty(ei.shape, dtype=ei.dtype, pin_memory=True)
                host.copy_(ei)
                ei = host
            ea, na = ea.cpu(), na.cpu()
        out = []
        for b in range(self.num_graphs):
            n0, n1 = int(self.host.node_ptr[b]), int(self.host.node_ptr[b + 1])
            out.append(SimpleNamespace(edge_index=ei[:, sl[b]:sl[b + 1]], num_nodes=n1 - n0,
                                       edge_attr=ea[sl[b]:sl[b + 1]], node_attr=na[n0:n1]))
        return out

    def to_networkx(self):
        """list of undirected networkx graphs (`pyg.utils.to_networkx(d, to_undirected=True)` per graph,
        experiment.py:70-75): one host copy of the edge list, numpy views per graph."""
        import networkx as nx
        out = []
        for d in self.to_data_list():
            G = nx.Graph()
            G.add_nodes_from(range(d.num_nodes))
            G.add_edges_from(d.edge_index.numpy().T.tolist())
            out.append(G)
        return out


def to_graph_batch(ref_batch, device='cuda', gid_base=0):
    """A batch in the REFERENCE's layout -> GraphBatch (host side, once per batch).

    `ref_batch` is what `collate_fn(preprocess(g, degree=True))` builds (datasets/data_utils.py:43-75) and what
    `train_fn` hands to `elbo_bpd` (experiment.py:28-33): `nodes_per_graph` int64[B], `degree` [Nn] (int64 from
    `preprocess`, fp32 from the empty-graph sampler; optional), and the x0 edges either as the dense pair state
    `full_edge_index` int64[2,P] + `full_edge_attr` int64[P] (the fields the reference path reads) or, when those
    are absent, as PyG's `edge_index` int64[2,2E] (both directions, global node ids).  Neither present: an empty
    state (a sampling batch).  The dense fields are read once here and never stored."""
    npg = torch.as_tensor(ref_batch.nodes_per_graph).cpu().long()
    B = int(npg.numel())
    ptr = torch.zeros(B + 1, dtype=torch.long)
    ptr[1:] = torch.cumsum(npg, 0)
    deg, ddt = None, None
    d = getattr(ref_batch, 'degree', None)
    if d is not None:
        d = torch.as_tensor(d).cpu()
        if int(d.numel()) != int(ptr[-1]):
            raise ValueError(f'degree has {int(d.numel())} entries for {int(ptr[-1])} nodes')
        ddt = np.float32 if d.dtype.is_floating_point else np.int64
        deg = [d[int(ptr[b]):int(ptr[b + 1])].numpy() for b in range(B)]
    fei, fea = getattr(ref_batch, 'full_edge_index', None), getattr(ref_batch, 'full_edge_attr', None)
    ei = getattr(ref_batch, 'edge_index', None)
    ge = None
    if fei is not None and fea is not None:
        fei, fea = torch.as_tensor(fei).cpu(), torch.as_tensor(fea).cpu()
        if fei.shape[1] != int((npg * (npg - 1) // 2).sum()) or fea.shape[0] != fei.shape[1]:
            raise ValueError('full_edge_index / full_edge_attr do not have sum n(n-1)/2 entries')
        ge = fei[:, fea.reshape(-1) > 0].t()
    elif ei is not None:
        ei = torch.as_tensor(ei).cpu()
        ge = ei[:, ei[0] < ei[1]].t()
    out = GraphBatch(npg.tolist(), degree=deg, device=device, gid_base=gid_base, degree_dtype=ddt)
    if ge is not None:
        ge = ge.contiguous().numpy()
        if ge.shape[0]:
            g0, g1 = np.searchsorted(ptr.numpy(), ge[:, 0], 'right'), np.searchsorted(ptr.numpy(), ge[:, 1], 'right')
            if ge.min() < 0 or ge.max() >= int(ptr[-1]) or not np.array_equal(g0, g1):
                raise ValueError('an edge leaves its graph (node ids must be global and within one graph)')
        out.set_edges(out.adj, ge)
        if deg is None:
            dg = torch.empty(out.num_nodes, dtype=torch.int32, device=out.device)
            with out._dev():
                capi.check(capi.lib.edgehip_degree_from_adj(C_byref(out.layout), out.adj.data_ptr(), dg.data_ptr(),
                                                            capi.stream_ptr()), 'degree_from_adj')
            out.degree_i32, out.degree = dg, dg.long()
    return out


def C_byref(x):
    import ctypes
    return ctypes.byref(x)
