"""GPU: the corners of BASELINE.json's configurations that no other file reaches.

  * config 5 (SBM, 20 000 nodes): a whole training step - forward, backward, clip, Adam - on one graph, its loss
    against the no-grad inference kernels on the same (t, noise), and the argument check that refuses layouts whose
    32-bit pair-tile counters could wrap;
  * config 2 (Ego shape, n 50-400, T 128, batch 8): a whole SAMPLING trajectory against the oracle run live in
    lockstep on the same uniforms;
  * config 3 (polblogs, 64 generations): graph g of the 64-graph / two-stream / device-RNG run equals the same graph
    sampled alone with gid_base = g (the B = 1 case is the one the reference golden trajectory pins);
  * the adapter from the reference's batch layout (datasets/data_utils.py:43-75) through the boundary: what a
    train.py with one changed import would run.
"""
import ctypes as C

import numpy as np
import pytest
import torch

import cases
import helpers as H
from oracle import edge_oracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-4


# ---------------------------------------------------------------------------------------------------------------
# config 5
# ---------------------------------------------------------------------------------------------------------------
def test_sbm_20k_full_training_step():
    """One SBM graph of 20 000 nodes (294 k edges, P = 2e8 pairs; diffusion_binomial_active.py:228-264 + autograd +
    experiment.py:33-37): `edgehip.train_step` (fused training form, gradient arena, clip, Adam) for three steps at
    different t.  Every step: the loss equals the fused INFERENCE kernels' loss on a fresh batch with the same
    (t, noise seed) and the weights of that moment to 1e-4, all gradients are finite and not all zero, the weights
    move.  The 20 000-row layout runs the backward's int64 row / pair indexing at P = 2e8."""
    import edgehip
    from edgehip import synth
    n, edges = synth.sbm_graph(20, 1000, 0.02, 5e-4, 0)
    d0 = synth.degrees(n, edges)
    T = 512
    model = H.build_model(orc.random_tgnn_params(4), T, int(d0.max()))
    model.train()
    tb = edgehip.GraphBatch([n], degree=[d0], edges=[edges], device='cuda', degree_dtype=np.int64)
    assert tb.num_pairs == n * (n - 1) // 2 > 1.9e8
    opt = edgehip.make_optimizer(model)
    state = {}
    w0 = model._denoise_fn.node_in[0].weight.detach().clone()
    pt = torch.full((1,), 1.0 / T)
    for step, tt in enumerate((60, 300, 5)):
        t = torch.tensor([tt])
        seed = 4100 + step
        model.eval()
        model.rng_seed = seed
        with torch.no_grad():
            fresh = edgehip.GraphBatch([n], degree=[d0], edges=[edges], device='cuda', degree_dtype=np.int64)
            ref = model.log_prob(fresh, t, pt)
            A_ref = int(fresh._state['aptr'][-1].item())
        model.train()
        model.rng_seed = seed
        want = {}
        bpd = edgehip.train_step(model, tb, opt, clip_value=1.0, state=state, t=t, pt=pt, want=want)
        got = -(want['kl'] / want['pt'] + want['kl_prior'])
        np.testing.assert_allclose(got.cpu().numpy(), ref.cpu().numpy(), rtol=TOL, atol=1e-9, err_msg=f'step {step}')
        assert int(tb._state['aptr'][-1].item()) == A_ref and (A_ref > 100 or tt > 250)
        assert np.isfinite(float(bpd))
        flat = state['arena'].flat
        assert bool(torch.isfinite(flat).all())
        if A_ref:
            assert float(flat.abs().max()) > 0
    assert float((model._denoise_fn.node_in[0].weight.detach() - w0).abs().max()) > 0


def test_layouts_whose_pair_tile_counters_could_wrap_are_refused():
    """tptr / counts->pair_tiles are int32 (include/edgehip.h): 86 graphs of 20 000 nodes could hold 2^29 pair tiles,
    and the offsets entry point - hence the sampler and the loss - refuses that layout before launching anything;
    85 pass."""
    import edgehip
    from edgehip import capi
    n = 20000
    per = (n * (n - 1) // 2 + 31) // 32 + 1
    fits = (1 << 29) // per
    assert fits == 85
    for B, ok in ((fits, True), (fits + 1, False)):
        L = capi.Layout()
        L.num_graphs, L.max_nodes, L.num_nodes = B, n, 0
        assert capi.lib.edgehip_layout_pair_tile_bound(C.byref(L)) == B * per
        z = torch.zeros(B + 2, dtype=torch.int64, device='cuda')        # act_count = 0: nothing active
        cnt = torch.zeros(8, dtype=torch.int32, device='cuda')
        rc = capi.lib.edgehip_active_offsets(C.byref(L), z.data_ptr(), z.clone().data_ptr(), z.clone().data_ptr(),
                                             z.clone().data_ptr(), cnt.data_ptr(), capi.stream_ptr())
        torch.cuda.synchronize()
        assert rc == (capi.OK if ok else capi.ERR_TOO_LARGE)
    with pytest.raises(capi.EdgeHipError, match='fixed capacity'):
        capi.check(capi.ERR_TOO_LARGE, 'active_offsets')


# ---------------------------------------------------------------------------------------------------------------
# lockstep replay against the live oracle
# ---------------------------------------------------------------------------------------------------------------
def _lockstep_trajectory(gs, T, params, useed):
    """Reverse trajectory of the oracle (diffusion_base.py:190-230) and of the op-by-op GPU path from the same state
    on the same uniforms, step by step.  Degrees-derived active lists must be equal at every step.  Sampled bits are
    exact given logits + uniforms, and the logits are fp32 network outputs that differ in the last bits, so a
    decision whose margin is below that noise may differ: such a step is recorded (margin asserted < 1e-4), the GPU
    state is re-synchronised to the oracle's and the replay goes on.  Returns (decisions, differing decisions, final
    oracle batch, GPU batch)."""
    from edgehip import synth
    dmax = cases.max_degree_of(gs)
    Nn = sum(x['n'] for x in gs)
    rng = np.random.default_rng(useed)
    o = orc.EdgeOracle(params, T, dmax)
    ob = orc.make_batch([dict(n=x['n'], edges=None, degree=synth.degrees(x['n'], x['edges'])) for x in gs])
    orc.prepare_data_for_sampling(o.sched, ob, orc.CallbackTape(lambda tag, shape: cases.uniforms(rng, *shape)))
    model = H.build_model(params, T, dmax)
    b = H.graph_batch(gs, with_edges=False)
    b.adj.zero_()
    fei = ob.full_edge_index
    decisions, diffs = 0, []
    for t in reversed(range(T)):
        drawn = []

        def take(tag, shape):
            drawn.append(cases.uniforms(rng, *shape))
            return drawn[-1]
        rec = {}
        with torch.no_grad():
            ln, le = o.p_sample(ob, torch.full((Nn,), t, dtype=torch.long), orc.CallbackTape(take), rec)
        ob.log_full_edge_attr_t, ob.log_node_attr_t = le, ln
        A = len(rec['active_edge_indices'])
        assert len(drawn) == (3 if A else 1)
        want = {}
        model.p_sample(b, t, drawn[0].cuda(), drawn[2].cuda() if A else None, None, None, want)
        ani, aei = model.materialize_actives(b)
        assert np.array_equal(ani.cpu().numpy(), rec['active_node_indices'].numpy()), f't={t}'
        assert np.array_equal(aei.cpu().numpy(), rec['active_edge_indices'].numpy()), f't={t}'
        decisions += A
        ref_bits = le.argmax(1)
        if A:
            H.check_logits(want['logits'], rec['logits'], TOL, what=f'logits t={t}')
            wrong = (want['bits'].cpu().long() != rec['sampled_bits']).nonzero(as_tuple=True)[0]
            if len(wrong):
                lp = torch.log_softmax(want['logits'].cpu().double(), 1)
                gum = orc.gumbel(drawn[2].double())
                margin = (gum[:, 1] + lp[:, 1]) - (gum[:, 0] + lp[:, 0])
                for k in wrong.tolist():
                    diffs.append(dict(t=t, k=k, margin=float(margin[k])))
                idx = ref_bits.nonzero(as_tuple=True)[0]
                b.adj.zero_()
                if len(idx):
                    b.set_edges(b.adj, fei[:, idx].t().contiguous().numpy())
        assert torch.equal(b.pair_bits().cpu(), ref_bits), f't={t}'
    bits = ob.log_full_edge_attr_t.argmax(-1)
    is_edge = bits.nonzero(as_tuple=True)[0]
    ob.edge_index = fei[:, is_edge]
    es = orc.scatter_sum(torch.ones_like(ob.edge_index[0]), ob.batch[ob.edge_index[0]], ob.num_graphs)
    ob.edge_slice = torch.cumsum(torch.nn.functional.pad(es, (1, 0)), 0)
    model._finish_sample(b)
    return decisions, diffs, ob, b


def test_config2_ego_T128_batch8_sampling_trajectory_vs_oracle():
    """SURVEY 8d config 2, the SAMPLING half (diffusion_base.py:199-207): eight ego-net shaped graphs (50-400 nodes,
    power-law degrees), T = 128, every reverse step against the oracle run live; the final edge lists are equal."""
    from edgehip import synth
    gs = [dict(n=n, edges=e) for n, e in synth.ego_like(8, seed=21)]
    assert min(x['n'] for x in gs) >= 50 and max(x['n'] for x in gs) <= 400
    decisions, diffs, ob, b = _lockstep_trajectory(gs, 128, orc.random_tgnn_params(17), 3131)
    assert decisions > 2000, decisions
    assert len(diffs) <= 4 and all(abs(d['margin']) < 1e-4 for d in diffs), diffs
    assert b.edge_index.shape[1] > 100
    assert np.array_equal(b.edge_index.cpu().numpy(), ob.edge_index.numpy())
    assert np.array_equal(b._slice_dict['edge_index'].cpu().numpy(), ob.edge_slice.numpy())


# ---------------------------------------------------------------------------------------------------------------
# config 3: a graph inside the bench's batch == the graph alone
# ---------------------------------------------------------------------------------------------------------------
def test_polblogs_graph_inside_the_64_graph_two_stream_run_equals_the_graph_alone():
    """BASELINE config 3 runs 64 polblogs generations per call (bench.py: two blocks of 32 on two streams, device
    RNG); the reference golden trajectory pins B = 1.  Graph g of the B = 64 run must be that same graph sampled
    alone with gid_base = g: compared for the first / last graph of each block and one in the middle, T shortened."""
    import edgehip
    from edgehip import synth
    n, e = synth.polblogs_like(0)
    d = synth.degrees(n, e)
    T, seed, B = 48, 20240, 64
    model = H.build_model(orc.random_tgnn_params(3), T, int(d.max()))
    big = edgehip.GraphBatch([n] * B, degree=[d] * B, device='cuda')
    assert model.default_parts(B) == 2
    model.sample_batch(big, seed)                        # default: two streams
    sl = big._slice_dict['edge_index'].cpu().tolist()
    assert sl[-1] > 64 * 20
    for g in (0, 17, 31, 32, 63):
        one = edgehip.GraphBatch([n], degree=[d], device='cuda', gid_base=g)
        model.sample_batch(one, seed)
        mine = big.edge_index[:, sl[g]:sl[g + 1]] - g * n
        assert one.edge_index.shape[1] > 5
        assert torch.equal(one.edge_index, mine), f'graph {g}'
    # and the graphs are not copies of each other (the uniforms are keyed by the global graph id)
    a0 = big.edge_index[:, sl[0]:sl[1]]
    a1 = big.edge_index[:, sl[1]:sl[2]] - n
    assert a0.shape != a1.shape or not torch.equal(a0, a1)


# ---------------------------------------------------------------------------------------------------------------
# the adapter from the reference's batch layout
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('name,case', [('loss_ragged', 'ragged'), ('loss_polblogs1', 'polblogs1')])
@pytest.mark.parametrize('fields', ['full_edge', 'edge_index'])
def test_reference_layout_batch_through_the_adapter_gives_the_golden_loss(name, case, fields):
    """`edgehip.to_graph_batch` on a batch in the reference's layout (collate_fn of preprocess'ed graphs,
    datasets/data_utils.py:43-75: full_edge_index / full_edge_attr / degree / nodes_per_graph / batch, or PyG's
    bidirectional edge_index) -> `model.log_prob` == the reference's golden -vb: what experiment.py:28-33 runs when
    train.py's import of get_model is the only change."""
    import edgehip
    g = H.golden(name + '.npz')
    builder, T, wseed, useed = cases.CASES[case]
    graphs = builder()
    ref_batch = orc.make_batch(graphs, degree_dtype=torch.int64)       # the reference-layout fields
    if fields == 'edge_index':
        ei = ref_batch.full_edge_index[:, ref_batch.full_edge_attr > 0]
        ref_batch.edge_index = torch.cat([ei, ei.flip(0)], 1)[:, torch.randperm(2 * ei.shape[1], generator=torch.Generator().manual_seed(1))]
        del ref_batch.full_edge_index, ref_batch.full_edge_attr
    b = edgehip.to_graph_batch(ref_batch)
    assert b.degree.dtype == torch.int64 and torch.equal(b.degree.cpu(), ref_batch.degree)
    assert torch.equal(b.nodes_per_graph.cpu(), ref_batch.nodes_per_graph)
    assert torch.equal(b.edges_per_graph.cpu(), ref_batch.edges_per_graph) and torch.equal(b.batch.cpu(), ref_batch.batch)
    rng = np.random.default_rng(useed + 1)
    Nn = sum(x['n'] for x in graphs)
    P = sum(x['n'] * (x['n'] - 1) // 2 for x in graphs)
    us = [cases.uniforms(rng, Nn, 2), cases.uniforms(rng, P, 2), cases.uniforms(rng, Nn, 2), cases.uniforms(rng, P, 2)]
    model = H.build_model(orc.random_tgnn_params(wseed), T, int(g['max_degree']))
    model.train()
    with torch.no_grad():
        nv = model.log_prob(b, torch.from_numpy(g['t']), torch.from_numpy(g['pt']), us[1].cuda(), us[3].cuda())
    np.testing.assert_allclose(nv.cpu().numpy(), g['neg_vb'], rtol=TOL, atol=1e-12)
    assert b.num_entries == int(g['num_entries'])
    assert np.array_equal(b.pair_bits(b._state['adj_t']).nonzero(as_tuple=True)[0].cpu().numpy(), g['xt'])


def test_adapter_rejects_inconsistent_reference_batches_and_builds_sampling_batches():
    import edgehip
    from types import SimpleNamespace as NS
    npg = torch.tensor([3, 4])
    with pytest.raises(ValueError):
        edgehip.to_graph_batch(NS(nodes_per_graph=npg, degree=torch.zeros(6)))
    with pytest.raises(ValueError):                                   # an edge between graph 0 and graph 1
        edgehip.to_graph_batch(NS(nodes_per_graph=npg, edge_index=torch.tensor([[0, 5], [5, 0]])))
    with pytest.raises(ValueError):
        edgehip.to_graph_batch(NS(nodes_per_graph=npg, full_edge_index=torch.zeros((2, 5), dtype=torch.long),
                                  full_edge_attr=torch.zeros(5, dtype=torch.long)))
    # a sampling batch as the reference's empty-graph sampler builds it: fp32 degree, no edges
    b = edgehip.to_graph_batch(NS(nodes_per_graph=npg, degree=torch.tensor([1., 1, 0, 2, 1, 1, 0])))
    assert b.degree.dtype == torch.float32 and int(b.adj.abs().sum()) == 0 and b.num_pairs == 3 + 6
    # degrees derived from the edges when the batch carries none
    b = edgehip.to_graph_batch(NS(nodes_per_graph=npg, edge_index=torch.tensor([[0, 1, 3, 4], [1, 0, 4, 3]])))
    assert b.degree_i32.cpu().tolist() == [1, 1, 0, 1, 1, 0, 0]
