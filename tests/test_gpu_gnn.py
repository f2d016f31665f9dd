"""GPU: GEMM, network forward, reverse step, loss and sampling trajectories through the C-ABI.

Tolerances (BASELINE.json north_star): integer / index results bit-exact; GNN logits and VB/CE
loss within 1e-4 relative in fp32.  "Relative" is max|gpu - ref| / max|ref| over the tensor."""
import ctypes as C

import numpy as np
import pytest
import torch

import cases
import helpers as H
from oracle import edge_oracle as orc

pytestmark = pytest.mark.gpu
byref = C.byref
TOL = 1e-4


@pytest.mark.parametrize('M,K,N,act', [(1, 64, 32, 0), (33, 64, 64, 1), (1222, 64, 512, 0), (4999, 576, 64, 1),
                                       (257, 256, 128, 1), (64, 128, 192, 0)])
def test_linear_fwd_vs_torch_fp32(M, K, N, act):
    from edgehip import capi
    g = torch.Generator().manual_seed(M * 7 + K + N)
    X = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / K ** 0.5
    bias = torch.randn(N, generator=g)
    Y = torch.empty(M, N, device='cuda')
    Xd, Wd, bd = X.cuda(), W.cuda(), bias.cuda()
    capi.check(capi.lib.edgehip_linear_fwd(Xd.data_ptr(), M, K, Wd.data_ptr(), bd.data_ptr(), N, act, Y.data_ptr(),
                                           capi.stream_ptr()))
    ref = torch.nn.functional.linear(X.double(), W.double(), bias.double())
    if act:
        ref = torch.nn.functional.silu(ref)
    assert H.rel_err(Y, ref) < 2e-6
    # A = I with an asymmetric B catches a transposed C write (guide section 3)
    if K == 64 and N == 64:
        I = torch.eye(64, device='cuda')[:M % 64 + 1].contiguous()
        Y2 = torch.empty(I.shape[0], N, device='cuda')
        capi.check(capi.lib.edgehip_linear_fwd(I.data_ptr(), I.shape[0], K, Wd.data_ptr(), None, N, 0, Y2.data_ptr(),
                                               capi.stream_ptr()))
        assert torch.equal(Y2.cpu(), W.t()[:I.shape[0]].contiguous())


@pytest.mark.parametrize('case,t_step', [('toy', 5), ('ragged', 9), ('polblogs1', 100)])
def test_p_sample_step_vs_reference_golden(case, t_step):
    """One reverse step on the reference's inputs: degrees, active sets, sampled edges bit-exact;
    logits within 1e-4 (golden produced by the reference's own p_sample)."""
    g = H.golden(f'step_{case}_t{t_step}.npz')
    builder, T, wseed, useed = cases.CASES[case]
    graphs = builder()
    rng = np.random.default_rng(useed)
    Nn = sum(x['n'] for x in graphs)
    P = sum(x['n'] * (x['n'] - 1) // 2 for x in graphs)
    cases.uniforms(rng, Nn, 2), cases.uniforms(rng, P, 2)
    u_act, u_node = cases.uniforms(rng, Nn, 2), cases.uniforms(rng, Nn, 2)
    model = H.build_model(orc.random_tgnn_params(wseed), T, int(g['max_degree']))
    b = H.graph_batch(graphs)
    H.set_state_from_pair_indices(b, g['xt_before'])
    want = {}
    model.p_sample(b, t_step, u_act.cuda(), lambda A: cases.uniforms(rng, A, 2).cuda(), None, None, want)
    ani, aei = model.materialize_actives(b)
    assert np.array_equal(b._state['deg_t'][:Nn].cpu().numpy().astype(np.int64), g['degree_t'])
    assert np.array_equal(ani.cpu().numpy(), g['active_node_indices'])
    assert np.array_equal(aei.cpu().numpy(), g['active_edge_indices'])
    H.check_logits(want['logits'], g['logits'], TOL)
    assert np.array_equal(want['bits'].cpu().numpy().astype(np.int64), g['sampled_bits'])
    assert np.array_equal(b.pair_bits().nonzero(as_tuple=True)[0].cpu().numpy(), g['xt_after'])


@pytest.mark.parametrize('heads', [(8, 8, 8, 8, 1), (4, 2, 1), (1,), (2, 8)])
def test_network_forward_vs_oracle_head_configs(heads):
    """Denoiser contract: forward(batch, t_node, t_edge) -> (node passthrough, logits[A,2])."""
    graphs = cases.ragged_graphs(5)
    T = 48
    dmax = cases.max_degree_of(graphs)
    params = orc.random_tgnn_params(77, num_heads=heads)
    ob = orc.make_batch(graphs)
    s = orc.make_schedule(T)
    rng = np.random.default_rng(9)
    ob.log_node_attr = orc.index_to_log_onehot(ob.node_attr)
    ob.log_full_edge_attr = orc.index_to_log_onehot(ob.full_edge_attr)
    Nn, P = ob.num_nodes, ob.full_edge_attr.shape[0]
    t = 6
    ob.log_node_attr_t, ob.log_full_edge_attr_t = orc.q_sample(
        s, ob, torch.full((Nn,), t), torch.full((P,), t), orc.UniformTape([cases.uniforms(rng, Nn, 2), cases.uniforms(rng, P, 2)]))
    u = cases.uniforms(rng, Nn, 2)
    orc.p_sample_and_set_actives(s, ob, torch.full((Nn,), t), orc.UniformTape([u]))
    assert len(ob.active_edge_indices) > 50
    with torch.no_grad():
        ref = orc.tgnn_forward(params, ob, torch.full((Nn,), t), dmax, T, heads)
    import edgehip
    a = H.make_args(T, dmax, heads=heads)
    model = edgehip.get_model(a, None)
    model._denoise_fn.load_state_dict(params)
    model = model.cuda().eval()
    b = H.graph_batch(graphs)
    H.set_state_from_pair_indices(b, ob.log_full_edge_attr_t.argmax(1).nonzero(as_tuple=True)[0])
    model._p_sample_and_set_actives(b, t, u.cuda())
    _, logits = model._denoise_fn(b, torch.full((Nn,), t, device='cuda'), None)
    assert logits.shape == ref.shape
    H.check_logits(logits, ref, TOL)


@pytest.mark.parametrize('graph', ['ragged', 'polblogs'])
def test_network_forward_with_sharp_attention_vs_oracle(graph):
    """The attention walk keeps a softmax reference that moves only when a score leaves it 32 log2-units behind
    (csrc/gnn.hip: eh_walk_step).  Random-init weights give scores within a few units of each other, so the move is
    only taken at a row's first step; here query and key weights are scaled by 20 so that the scores of one row span
    up to ~140 units and the reference has to move in mid-row: on the polblogs-shaped graph the CPU oracle counts
    ~1 400 such moves in the 8-head layer (~260 of them on rows above 48 neighbours) and ~400 in the
    one-head layer.  The ragged graphs cover the short rows (spread up to ~90 units inside a row's first steps)."""
    heads = (8, 1)
    graphs = cases.ragged_graphs(5) if graph == 'ragged' else cases.polblogs_graph(0)
    T = 48
    dmax = cases.max_degree_of(graphs)
    params = orc.random_tgnn_params(78, num_heads=heads)
    for k in list(params):
        if '.lin_query.' in k or '.lin_key.' in k:
            params[k] = params[k] * 20.0
    ob = orc.make_batch(graphs)
    s = orc.make_schedule(T)
    rng = np.random.default_rng(10)
    ob.log_node_attr = orc.index_to_log_onehot(ob.node_attr)
    ob.log_full_edge_attr = orc.index_to_log_onehot(ob.full_edge_attr)
    Nn, P = ob.num_nodes, ob.full_edge_attr.shape[0]
    t = 3
    ob.log_node_attr_t, ob.log_full_edge_attr_t = orc.q_sample(
        s, ob, torch.full((Nn,), t), torch.full((P,), t), orc.UniformTape([cases.uniforms(rng, Nn, 2), cases.uniforms(rng, P, 2)]))
    u = cases.uniforms(rng, Nn, 2)
    orc.p_sample_and_set_actives(s, ob, torch.full((Nn,), t), orc.UniformTape([u]))
    assert len(ob.active_edge_indices) > 20
    with torch.no_grad():
        ref = orc.tgnn_forward(params, ob, torch.full((Nn,), t), dmax, T, heads)
    import edgehip
    a = H.make_args(T, dmax, heads=heads)
    model = edgehip.get_model(a, None)
    model._denoise_fn.load_state_dict(params)
    model = model.cuda().eval()
    b = H.graph_batch(graphs)
    H.set_state_from_pair_indices(b, ob.log_full_edge_attr_t.argmax(1).nonzero(as_tuple=True)[0])
    model._p_sample_and_set_actives(b, t, u.cuda())
    _, logits = model._denoise_fn(b, torch.full((Nn,), t, device='cuda'), None)
    assert logits.shape == ref.shape
    H.check_logits(logits, ref, TOL)


@pytest.mark.parametrize('differentiable', [False, True], ids=['fused', 'train_ops'])
@pytest.mark.parametrize('name,case', [('loss_toy', 'toy'), ('loss_toy_mid', 'toy'), ('loss_ragged', 'ragged'),
                                       ('loss_polblogs1', 'polblogs1')])
def test_train_loss_vs_reference_golden(name, case, differentiable):
    """_train_loss against the reference's own outputs, through both forms of the forward: the fused
    inference kernels (torch.no_grad) and the op-by-op differentiable form (grad enabled)."""
    g = H.golden(name + '.npz')
    builder, T, wseed, useed = cases.CASES[case]
    graphs = builder()
    rng = np.random.default_rng(useed + 1)
    Nn = sum(x['n'] for x in graphs)
    P = sum(x['n'] * (x['n'] - 1) // 2 for x in graphs)
    us = [cases.uniforms(rng, Nn, 2), cases.uniforms(rng, P, 2), cases.uniforms(rng, Nn, 2), cases.uniforms(rng, P, 2)]
    model = H.build_model(orc.random_tgnn_params(wseed), T, int(g['max_degree']))
    model.train()
    b = H.graph_batch(graphs, degree_dtype=np.int64)
    want = {}
    with torch.set_grad_enabled(differentiable):
        nv = model.log_prob(b, torch.from_numpy(g['t']), torch.from_numpy(g['pt']), us[1].cuda(), us[3].cuda(), want)
    assert nv.requires_grad == differentiable
    nv = nv.detach()
    st = b._state
    assert np.array_equal(b.pair_bits(st['adj_tm1']).nonzero(as_tuple=True)[0].cpu().numpy(), g['xtm1'])
    assert np.array_equal(b.pair_bits(st['adj_t']).nonzero(as_tuple=True)[0].cpu().numpy(), g['xt'])
    ani, aei = model.materialize_actives(b)
    assert np.array_equal(ani.cpu().numpy(), g['active_node_indices'])
    assert np.array_equal(aei.cpu().numpy(), g['active_edge_indices'])
    if len(g['logits']):
        H.check_logits(want['logits'], g['logits'], TOL)
    np.testing.assert_allclose(want['kl'].cpu().numpy(), g['kl'], rtol=TOL, atol=1e-12)
    # kl_prior: the reference adds P identical fp32 terms one by one (scatter-add), which carries
    # ~1e-4 of rounding drift at P = 746k; the product uses count x value in fp64.  It is a <1e-5
    # share of the VB loss, whose 1e-4 bound is asserted on neg_vb below.
    np.testing.assert_allclose(want['kl_prior'].cpu().numpy(), g['kl_prior'], rtol=5e-4)
    # ... demonstrated: the same per-element fp32 values summed in fp64 give the product's number to 1e-6
    ob = orc.make_batch(graphs, degree_dtype=torch.int64)
    ob.log_node_attr = orc.index_to_log_onehot(ob.node_attr)
    ob.log_full_edge_attr = orc.index_to_log_onehot(ob.full_edge_attr)
    prior64 = H.kl_prior_fp64(orc.EdgeOracle(orc.random_tgnn_params(wseed), T, int(g['max_degree'])), ob).numpy()
    np.testing.assert_allclose(want['kl_prior'].cpu().numpy().astype(np.float64), prior64, rtol=1e-6)
    np.testing.assert_allclose(nv.cpu().numpy(), g['neg_vb'], rtol=TOL, atol=1e-12)
    import edgehip
    assert b.num_entries == int(g['num_entries'])
    np.testing.assert_allclose(model.Lt_count.cpu().numpy(), g['Lt_count'])
    np.testing.assert_allclose(model.Lt_history.cpu().numpy(), g['Lt_history'], rtol=2 * TOL)


@pytest.mark.parametrize('case', ['toy', 'ragged'])
def test_sample_trajectory_vs_reference_golden(case):
    """Whole reverse process with replayed uniforms: per-step active sets and the final edge list
    equal the reference's (diffusion_base.py:190-230)."""
    g = H.golden(f'traj_{case}.npz')
    builder, T, wseed, useed = cases.CASES[case]
    graphs = builder()
    rng = np.random.default_rng(useed + 2)
    Nn = sum(x['n'] for x in graphs)
    P = sum(x['n'] * (x['n'] - 1) // 2 for x in graphs)
    model = H.build_model(orc.random_tgnn_params(wseed), T, int(g['max_degree']))
    b = H.graph_batch(graphs, with_edges=False)
    cases.uniforms(rng, Nn, 2), cases.uniforms(rng, P, 2)          # prior draws (always the empty graph)
    for t in reversed(range(T)):
        u_act = cases.uniforms(rng, Nn, 2).cuda()

        def pair_u(A):
            cases.uniforms(rng, Nn, 2)                             # node draw, consumed only when A > 0
            return cases.uniforms(rng, A, 2).cuda()
        model.p_sample(b, t, u_act, pair_u)
        ani, aei = model.materialize_actives(b)
        assert np.array_equal(ani.cpu().numpy(), g[f'K_{t}']), t
        assert np.array_equal(aei.cpu().numpy(), g[f'A_{t}']), t
        assert np.array_equal(b.pair_bits().nonzero(as_tuple=True)[0].cpu().numpy(), g[f'state_{t}']), t
    model._finish_sample(b)
    assert np.array_equal(b.edge_index.cpu().numpy(), g['edge_index'])
    assert np.array_equal(b._slice_dict['edge_index'].cpu().numpy(), g['edge_slice'])
    dl = b.to_data_list()
    assert [d.edge_index.shape[1] for d in dl] == np.diff(g['edge_slice']).tolist()
    _check_consumer_format(b, g['edge_index'], g['edge_slice'])


def _check_consumer_format(b, edge_index, edge_slice):
    """What the reference's callers do with the result (experiment.py:70-75, evaluate.py:51-58): to_data_list(),
    per-graph local ids, networkx hand-off; here as views of one tensor, on the host (one pinned D2H copy) or on the
    device, never one copy per graph."""
    import networkx as nx
    node_ptr = b.host.node_ptr
    assert np.array_equal(b._inc_dict['edge_index'].cpu().numpy(), node_ptr[:-1])         # diffusion_base.py:228
    for dev in (None, 'cuda'):
        dl = b.to_data_list(device=dev)
        assert len(dl) == b.num_graphs
        base = None
        for k, d in enumerate(dl):
            want = edge_index[:, edge_slice[k]:edge_slice[k + 1]] - node_ptr[k]
            assert d.edge_index.is_cuda == (dev == 'cuda')
            assert np.array_equal(d.edge_index.cpu().numpy(), want)
            assert d.num_nodes == node_ptr[k + 1] - node_ptr[k] and d.node_attr.shape[0] == d.num_nodes
            assert d.edge_attr.shape[0] == want.shape[1] and bool((d.edge_attr == 1).all())
            if want.shape[1]:
                st = d.edge_index.untyped_storage().data_ptr()
                base = st if base is None else base
                assert st == base                                                           # views of ONE tensor
        if dev is None and base is not None:
            assert next(d for d in dl if d.edge_index.shape[1]).edge_index.is_pinned()
    gs = b.to_networkx()
    for k, G in enumerate(gs):
        want = edge_index[:, edge_slice[k]:edge_slice[k + 1]] - node_ptr[k]
        H2 = nx.Graph()
        H2.add_nodes_from(range(int(node_ptr[k + 1] - node_ptr[k])))
        H2.add_edges_from(want.T.tolist())
        assert G.number_of_nodes() == H2.number_of_nodes() and set(map(frozenset, G.edges)) == set(map(frozenset, H2.edges))


def test_fused_sampler_matches_stepwise_and_is_shard_invariant():
    """edgehip_sample_step (fused, device RNG, no host sync) == the op-by-op path with the same
    Philox uniforms, and graph b's result does not depend on which rank/batch holds it."""
    from edgehip import capi
    graphs = cases.ragged_graphs(3)
    T, seed = 48, 99
    dmax = cases.max_degree_of(graphs)
    model = H.build_model(orc.random_tgnn_params(5), T, dmax)
    b1 = H.graph_batch(graphs, with_edges=False)
    model.sample_batch(b1, seed)
    b2 = H.graph_batch(graphs, with_edges=False)
    b2.adj.zero_()
    for t in reversed(range(T)):
        model.p_sample(b2, t, None, None, capi.Rng(seed))
    model._finish_sample(b2)
    assert torch.equal(b1.edge_index, b2.edge_index)
    assert b1.edge_index.shape[1] > 20
    # shard invariance: graphs 3.. alone with gid_base=3
    b3 = H.graph_batch(graphs[3:], with_edges=False, gid_base=3)
    model.sample_batch(b3, seed)
    sl = b1._slice_dict['edge_index'].tolist()
    off = int(b1.host.node_ptr[3])
    assert torch.equal(b3.edge_index, b1.edge_index[:, sl[3]:] - off)


@pytest.mark.parametrize('parts', [2, 3, 5])
def test_sampler_on_sub_batch_streams_is_bit_identical(parts):
    """sample_batch(parts=k): contiguous blocks of graphs on separate HIP streams sharing the batch's state;
    graphs do not interact and the uniforms are keyed by global graph id, so the edge lists are those of the
    single-stream run, bit for bit (ragged batch incl. 1-node, 2-node and empty graphs, block borders in between)."""
    graphs = cases.ragged_graphs(3) + cases.ragged_graphs(4) + cases.ragged_graphs(5)[:5]
    T, seed = 40, 1234
    dmax = cases.max_degree_of(graphs)
    model = H.build_model(orc.random_tgnn_params(5), T, dmax)
    b1 = H.graph_batch(graphs, with_edges=False, gid_base=7)
    model.sample_batch(b1, seed, parts=1)
    bk = H.graph_batch(graphs, with_edges=False, gid_base=7)
    model.sample_batch(bk, seed, parts=parts)
    assert b1.edge_index.shape[1] > 50
    assert torch.equal(b1.edge_index, bk.edge_index)
    assert torch.equal(b1._slice_dict['edge_index'], bk._slice_dict['edge_index'])
    # a second run on the same batch object reuses the cached sub-batches and streams
    model.sample_batch(bk, seed + 1, parts=parts)
    model.sample_batch(b1, seed + 1, parts=1)
    assert torch.equal(b1.edge_index, bk.edge_index)
    assert model.default_parts(64) == 2 and model.default_parts(7) == 1 and model.default_parts(16) == 2
    # new target degrees written into the same batch object: the cached blocks (which copy d0) are rebuilt
    halved = torch.div(bk.degree_i32, 2, rounding_mode='floor')
    for b in (b1, bk):
        b.degree_i32.copy_(halved)
        b.degree.copy_(halved.to(b.degree.dtype))
    model.sample_batch(bk, seed + 2, parts=parts)
    model.sample_batch(b1, seed + 2, parts=1)
    assert torch.equal(b1.edge_index, bk.edge_index)


def test_sample_api_polblogs_shape():
    """model.sample(n) end to end at polblogs size (T shortened to 64 for test time): symmetric simple
    graph, ascending edge list, deterministic under a fixed seed."""
    import edgehip
    from edgehip import synth
    n, e = synth.polblogs_like(0)
    sampler = edgehip.EmpiricalEmptyGraphGenerator([(n, e)], degree=True, device='cuda')
    d = synth.degrees(n, e)
    model = H.build_model(orc.random_tgnn_params(3), 64, int(d.max()), sampler)
    out = model.sample(4, seed=7)
    ei, sl = out.edge_index, out._slice_dict['edge_index']
    assert out.num_graphs == 4 and out.num_nodes == 4 * n
    assert bool((ei[0] < ei[1]).all())
    key = ei[0] * (4 * n) + ei[1]
    assert bool((key[1:] > key[:-1]).all())                       # ascending pair order, no duplicates
    assert bool((out.batch[ei[0]] == out.batch[ei[1]]).all())
    assert int(sl[-1]) == ei.shape[1]
    out2 = model.sample_batch(edgehip.GraphBatch([n] * 4, degree=[d] * 4, device='cuda'), seed=7)
    assert torch.equal(out2.edge_index, ei)


@pytest.mark.parametrize('state', ['empty', 'sparse', 'dense'])
def test_live_row_compaction_matches_full_computation(state):
    """Isolated nodes that share (d0, active flag) are computed once (k_compact).  The same forward with the
    compaction off (h_out requested: every node is a row) must give the same per-active-node vectors."""
    from edgehip import capi
    from edgehip.model import _workspace
    graphs = cases.ragged_graphs(6)
    T = 48
    dmax = cases.max_degree_of(graphs)
    model = H.build_model(orc.random_tgnn_params(8), T, dmax)
    b = H.graph_batch(graphs)
    rng = np.random.default_rng(4)
    keep = b.pair_bits().nonzero(as_tuple=True)[0].cpu().numpy()
    frac = dict(empty=0.0, sparse=0.15, dense=0.6)[state]
    H.set_state_from_pair_indices(b, keep[rng.random(len(keep)) < frac])
    Nn = b.num_nodes
    model._p_sample_and_set_actives(b, 2, cases.uniforms(rng, Nn, 2).cuda())
    st, L = b._state, b.layout
    K = int(st['kptr'][-1].item())
    assert K > 6
    P, folded = model._denoise_fn.c_params()
    ws = _workspace(b, P)
    outs = []
    for full in (False, True):
        node_repr = torch.zeros((Nn + 32, 128), dtype=torch.float32, device='cuda')
        h_out = torch.zeros((Nn, 64), dtype=torch.float32, device='cuda') if full else None
        capi.check(capi.lib.edgehip_gnn_forward(
            byref(L), byref(P), folded.data_ptr(), b.adj.data_ptr(), st['deg_t'].data_ptr(), b.degree_i32.data_ptr(),
            st['t_graph'].data_ptr(), 0, st['flags'].data_ptr(), st['act_nodes'].data_ptr(), st['kptr'].data_ptr(),
            st['counts'].data_ptr(), node_repr.data_ptr(), capi.ptr(h_out), ws.data_ptr(), ws.numel(), capi.stream_ptr()))
        outs.append(node_repr[:K].clone())
    kp = st['kptr'].cpu().numpy()
    rows = np.concatenate([np.arange(kp[g], kp[g + 1]) for g in range(b.num_graphs) if kp[g + 1] - kp[g] >= 2])
    assert len(rows) > 4
    assert H.rel_err(outs[0][rows], outs[1][rows]) < 2e-6


def test_one_model_samples_batches_of_growing_and_shrinking_max_degree():
    """The posterior table is sized by the largest target degree seen so far and rebuilt when a batch exceeds it:
    a model that sampled a low-degree batch, then a high-degree one, then the first again gives on each what a fresh
    model gives (bit for bit), whatever table it holds at that moment."""
    low = cases.ragged_graphs(3)
    high = cases.ragged_graphs(9) + cases.ragged_graphs(4)
    T = 40
    dmax = max(cases.max_degree_of(low), cases.max_degree_of(high))
    assert cases.max_degree_of(low) != cases.max_degree_of(high)
    params = orc.random_tgnn_params(8)
    model = H.build_model(params, T, dmax)
    for k, graphs in enumerate((low, high, low, high) if cases.max_degree_of(low) < cases.max_degree_of(high)
                               else (high, low, high, low)):
        b = H.graph_batch(graphs, with_edges=False, gid_base=2)
        model.sample_batch(b, 50 + k, parts=2)
        fb = H.graph_batch(graphs, with_edges=False, gid_base=2)
        H.build_model(params, T, dmax).sample_batch(fb, 50 + k, parts=1)
        assert torch.equal(b.edge_index, fb.edge_index), k
        assert torch.equal(b._slice_dict['edge_index'], fb._slice_dict['edge_index']), k


@pytest.mark.parametrize('schedule,t', [('cosine', 9), ('Tt1', 1)])
def test_reverse_step_and_loss_under_the_other_noise_schedules(schedule, t):
    """--noise_schedule cosine (the reference's default, model.py:22) and Tt1: the host tables are pinned by
    schedule.npz; this runs the DEVICE path under them - one reverse step (degrees, active sets, sampled bits
    bit-exact, logits 1e-4) and one training loss - against the oracle built with the same schedule."""
    graphs = cases.ragged_graphs(12)
    T = 48
    dmax = cases.max_degree_of(graphs)
    params = orc.random_tgnn_params(17)
    rng = np.random.default_rng(23)
    ob = orc.make_batch(graphs)
    Nn, P = ob.num_nodes, ob.full_edge_attr.shape[0]
    o = orc.EdgeOracle(params, T, dmax, schedule=schedule)
    ob.log_node_attr = orc.index_to_log_onehot(ob.node_attr)
    ob.log_full_edge_attr = orc.index_to_log_onehot(ob.full_edge_attr)
    ob.log_node_attr_t, ob.log_full_edge_attr_t = orc.q_sample(
        o.sched, ob, torch.full((Nn,), t), torch.full((P,), t),
        orc.UniformTape([cases.uniforms(rng, Nn, 2), cases.uniforms(rng, P, 2)]))
    start = ob.log_full_edge_attr_t.argmax(1).nonzero(as_tuple=True)[0]
    u_act, u_node = cases.uniforms(rng, Nn, 2), cases.uniforms(rng, Nn, 2)
    model = H.build_model(params, T, dmax, schedule=schedule)
    b = H.graph_batch(graphs)
    H.set_state_from_pair_indices(b, start)
    want, drawn = {}, {}

    def pair_u(A):
        drawn['u'] = cases.uniforms(rng, A, 2)
        return drawn['u'].cuda()
    model.p_sample(b, t, u_act.cuda(), pair_u, None, None, want)
    rec = {}
    with torch.no_grad():
        _, out = o.p_sample(ob, torch.full((Nn,), t), orc.UniformTape([u_act, u_node, drawn['u']]), rec)
    ani, aei = model.materialize_actives(b)
    assert len(aei) > 0
    assert torch.equal(ani.cpu(), rec['active_node_indices']) and torch.equal(aei.cpu(), rec['active_edge_indices'])
    H.check_logits(want['logits'], rec['logits'], TOL)
    assert torch.equal(b.pair_bits().cpu(), out.argmax(1))
    # training loss at mixed t under the same schedule
    ob2 = orc.make_batch(graphs, degree_dtype=torch.int64)
    ob2.log_node_attr = orc.index_to_log_onehot(ob2.node_attr)
    ob2.log_full_edge_attr = orc.index_to_log_onehot(ob2.full_edge_attr)
    us = [cases.uniforms(rng, Nn, 2), cases.uniforms(rng, P, 2), cases.uniforms(rng, Nn, 2), cases.uniforms(rng, P, 2)]
    tt = torch.tensor([0, 7, 47, 12, 1, 20])
    pt = torch.full((6,), 1.0 / T)
    rec2 = {}
    with torch.no_grad():
        nv_ref = o.loss(ob2, tt, pt, orc.UniformTape(us), training=False, record=rec2)
        nv = model.log_prob(H.graph_batch(graphs, degree_dtype=np.int64), tt, pt, us[1].cuda(), us[3].cuda())
    prior64 = H.kl_prior_fp64(o, ob2)
    nv64 = -(rec2['kl'].double() / pt.double() + prior64)
    np.testing.assert_allclose(nv.cpu().numpy().astype(np.float64), nv64.numpy(), rtol=TOL, atol=1e-12)
    np.testing.assert_allclose(nv.cpu().numpy(), nv_ref.numpy(), rtol=5e-4, atol=1e-12)
