"""GPU: edgehip_edge_ce_loss_static (the entry point that allocates nothing) against edgehip_edge_ce_loss, through
the C ABI, on two calls in a row - the second with a smaller input, so stale contents of the shared device array from
the first call are in place when it runs.  Collected last (file name) so a failure here hides no other test."""
import ctypes as C

import numpy as np
import pytest
import torch

import cases
import helpers as H
from oracle import edge_oracle as orc

pytestmark = pytest.mark.gpu
byref = C.byref


def _prepared(model, graphs, t_list, seed):
    """A batch with x_t / x_{t-1} drawn, the active set selected and the network run: everything both entry points read."""
    from edgehip import capi
    from edgehip.model import _state, _workspace
    b = H.graph_batch(graphs, degree_dtype=np.int64)
    L, dev = b.layout, b.device
    tabs = model._tables(dev, 0)
    st = _state(b)
    s = capi.stream_ptr()
    t_graph = torch.tensor(t_list, dtype=torch.int32, device=dev)
    adj_tm1, adj_t = b.new_adj(), b.new_adj()
    rng = capi.Rng(seed)
    capi.check(capi.lib.edgehip_q_sample_edges(byref(L), b.adj.data_ptr(), t_graph.data_ptr(), tabs['q_cum'].data_ptr(),
                                               tabs['q_one'].data_ptr(), None, None, byref(rng), capi.ptr(tabs['lut']),
                                               adj_tm1.data_ptr(), adj_t.data_ptr(), s), 'q_sample')
    deg_tm1 = torch.empty_like(st['deg_t'])
    capi.check(capi.lib.edgehip_degree_from_adj(byref(L), adj_tm1.data_ptr(), deg_tm1.data_ptr(), s), 'degree')
    capi.check(capi.lib.edgehip_degree_from_adj(byref(L), adj_t.data_ptr(), st['deg_t'].data_ptr(), s), 'degree')
    capi.check(capi.lib.edgehip_active_select_train(byref(L), deg_tm1.data_ptr(), st['deg_t'].data_ptr(), st['flags'].data_ptr(),
                                                    st['act_nodes'].data_ptr(), st['act_count'].data_ptr(), s), 'select')
    model._set_offsets(b)
    P, folded = model._denoise_fn.c_params()
    ws = _workspace(b, P)
    node_repr = torch.empty((b.num_nodes + 32, 128), dtype=torch.float32, device=dev)
    capi.check(capi.lib.edgehip_gnn_forward(byref(L), byref(P), folded.data_ptr(), adj_t.data_ptr(), st['deg_t'].data_ptr(),
                                            b.degree_i32.data_ptr(), t_graph.data_ptr(), 0, st['flags'].data_ptr(),
                                            st['act_nodes'].data_ptr(), st['kptr'].data_ptr(), st['counts'].data_ptr(),
                                            node_repr.data_ptr(), None, ws.data_ptr(), ws.numel(), s), 'gnn_forward')
    A, ntile = int(st['aptr'][-1].item()), int(st['tptr'][-1].item())
    return dict(b=b, L=L, P=P, folded=folded, st=st, adj_tm1=adj_tm1, node_repr=node_repr, A=A, ntile=ntile, keep=(ws, adj_t, t_graph))


def _both(x):
    from edgehip import capi
    b, L, P, st, dev = x['b'], x['L'], x['P'], x['st'], x['b'].device
    B, A = b.num_graphs, x['A']
    s = capi.stream_ptr()
    common = (byref(L), byref(P), x['folded'].data_ptr(), x['node_repr'].data_ptr(), st['act_nodes'].data_ptr(),
              st['kptr'].data_ptr(), st['aptr'].data_ptr(), st['tptr'].data_ptr(), st['counts'].data_ptr(),
              x['adj_tm1'].data_ptr())
    partial = torch.zeros(max(x['ntile'], 1), dtype=torch.float32, device=dev)
    kl_a = torch.full((B,), -1.0, dtype=torch.float32, device=dev)
    lg_a = torch.full((max(A, 1), 2), -7.0, dtype=torch.float32, device=dev)
    capi.check(capi.lib.edgehip_edge_ce_loss(*common, partial.data_ptr(), x['ntile'], kl_a.data_ptr(), lg_a.data_ptr(), A, s),
               'edge_ce_loss')
    kl_b = torch.full((B,), -2.0, dtype=torch.float32, device=dev)
    lg_b = torch.full((max(A, 1), 2), -9.0, dtype=torch.float32, device=dev)
    # a warm call first: the per-device address of the scratch array is resolved on the first call on a device
    capi.check(capi.lib.edgehip_edge_ce_loss_static(*common, kl_b.data_ptr(), lg_b.data_ptr(), A, s), 'edge_ce_loss_static')
    torch.cuda.synchronize()
    kl_b.fill_(-2.0); lg_b.fill_(-9.0)
    torch.cuda.synchronize()
    # (1) neither torch's allocator nor the DEVICE sees an allocation: free bytes as hipMemGetInfo reports them are
    #     unchanged across the call (torch's own statistics cannot see a hipMalloc inside the library)
    before = torch.cuda.memory_stats(dev)['allocation.all.allocated']
    free0 = torch.cuda.mem_get_info(dev)[0]
    capi.check(capi.lib.edgehip_edge_ce_loss_static(*common, kl_b.data_ptr(), lg_b.data_ptr(), A, s), 'edge_ce_loss_static')
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info(dev)[0] == free0
    assert torch.cuda.memory_stats(dev)['allocation.all.allocated'] == before
    assert int(st['counts'][4].item()) == 0            # status: no capacity flag
    # (2) the callers it is for: under HIP stream capture a hipMalloc / hipFree / synchronising call is an error (it
    #     would invalidate the capture and the library would return a launch error); the captured call replays and
    #     gives the same bits
    kl_c = torch.full((B,), -3.0, dtype=torch.float32, device=dev)
    lg_c = torch.full((max(A, 1), 2), -5.0, dtype=torch.float32, device=dev)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        rc = capi.lib.edgehip_edge_ce_loss_static(*common, kl_c.data_ptr(), lg_c.data_ptr(), A, capi.stream_ptr())
    assert rc == capi.OK
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(kl_c, kl_b) and torch.equal(lg_c, lg_b)
    return kl_a, kl_b, lg_a[:A], lg_b[:A]


def test_static_scratch_entry_point_equals_the_workspace_one_twice_in_a_row():
    from edgehip import capi
    assert capi.lib.edgehip_edge_ce_loss_static_max_tiles() == 1 << 21
    params = orc.random_tgnn_params(31)
    big = cases.polblogs_graph(0) + cases.polblogs_graph(1)                 # ~46 k pair tiles possible
    small = cases.ragged_graphs(12) + [dict(n=150, edges=cases.random_graph(np.random.default_rng(3), 150, 0.4))]
    dmax = max(cases.max_degree_of(big), cases.max_degree_of(small))
    model = H.build_model(params, 512, dmax)
    first = _prepared(model, big, [100, 60], seed=5)
    second = _prepared(model, small, [0, 7, 47, 12, 1, 20, 250], seed=6)
    assert first['ntile'] > second['ntile'] > 0 and first['A'] > 1000
    for x in (first, second):                                              # second call: smaller input, stale scratch
        kl_a, kl_b, lg_a, lg_b = _both(x)
        assert torch.equal(kl_a, kl_b), (kl_a, kl_b)                        # same kernels, same order of sums: bit-equal
        assert torch.equal(lg_a, lg_b)
        assert float(kl_a.abs().max()) > 0


def test_static_scratch_entry_point_rejects_layouts_beyond_its_capacity():
    from edgehip import capi
    import edgehip
    n = 12000                                 # 72 M pairs -> 2.25 M tiles > 2 Mi
    b = edgehip.GraphBatch([n], degree=[np.zeros(n)], device='cuda')
    z = torch.zeros(8, dtype=torch.int64, device='cuda')
    P = capi.GnnParams()
    rc = capi.lib.edgehip_edge_ce_loss_static(byref(b.layout), byref(P), z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(),
                                              z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), None, 0, None)
    assert rc == capi.ERR_TOO_LARGE
    with pytest.raises(capi.EdgeHipError, match='fixed capacity'):
        capi.check(rc, 'edge_ce_loss_static')
