"""GPU: bit-matrix state kernels through the C-ABI against the oracle (bit-exact)."""
import ctypes as C

import numpy as np
import pytest
import torch

import cases
import helpers as H
from oracle import edge_oracle as orc

pytestmark = pytest.mark.gpu
byref = C.byref


def _dense_bits(graphs):
    return orc.make_batch(graphs).full_edge_attr


@pytest.mark.parametrize('case', ['toy', 'ragged', 'polblogs1'])
def test_degree_and_roundtrip(case):
    from edgehip import capi
    graphs = cases.CASES[case][0]()
    ob = orc.make_batch(graphs)
    b = H.graph_batch(graphs)
    deg = torch.empty(b.num_nodes, dtype=torch.int32, device='cuda')
    capi.check(capi.lib.edgehip_degree_from_adj(byref(b.layout), b.adj.data_ptr(), deg.data_ptr(), capi.stream_ptr()))
    want = orc.compute_degree(ob.full_edge_attr, ob.full_edge_index, ob.num_nodes)
    assert torch.equal(deg.cpu().long(), want)
    # edge list -> bit matrix -> edge list round trip, in the reference's ascending pair order
    ei, sl = b.extract_edges()
    is_edge = ob.full_edge_attr.nonzero(as_tuple=True)[0]
    assert torch.equal(ei.cpu(), ob.full_edge_index[:, is_edge])
    cnt = orc.scatter_sum(torch.ones_like(is_edge), ob.batch[ob.full_edge_index[0, is_edge]], ob.num_graphs)
    assert torch.equal(sl.cpu(), torch.cumsum(torch.nn.functional.pad(cnt, (1, 0)), 0))
    assert torch.equal(b.pair_bits().cpu(), ob.full_edge_attr)


@pytest.mark.parametrize('case,t', [('toy', 5), ('ragged', 9), ('ragged', 40), ('polblogs1', 100), ('polblogs1', 400)])
def test_active_select_sample_and_enumerate(case, t):
    """a9/a10: posterior table + Gumbel draw + stable compaction + pair enumeration."""
    builder, T, wseed, useed = cases.CASES[case]
    graphs = builder()
    rng = np.random.default_rng(useed + 7 + t)
    ob = orc.make_batch(graphs)
    s = orc.make_schedule(T)
    # state: q-sample of the true graph at t (oracle), d0 = true degrees
    ob.log_node_attr = orc.index_to_log_onehot(ob.node_attr)
    ob.log_full_edge_attr = orc.index_to_log_onehot(ob.full_edge_attr)
    Nn, P = ob.num_nodes, ob.full_edge_attr.shape[0]
    tape = orc.UniformTape([cases.uniforms(rng, Nn, 2), cases.uniforms(rng, P, 2)])
    ob.log_node_attr_t, ob.log_full_edge_attr_t = orc.q_sample(s, ob, torch.full((Nn,), t), torch.full((P,), t), tape)
    u = cases.uniforms(rng, Nn, 2)
    orc.p_sample_and_set_actives(s, ob, torch.full((Nn,), t), orc.UniformTape([u]))

    model = H.build_model(orc.random_tgnn_params(wseed), T, cases.max_degree_of(graphs))
    b = H.graph_batch(graphs)
    H.set_state_from_pair_indices(b, ob.log_full_edge_attr_t.argmax(1).nonzero(as_tuple=True)[0])
    model._p_sample_and_set_actives(b, t, u.cuda())
    assert torch.equal(b._state['deg_t'][:Nn].cpu().long(), ob.degree_t)
    ani, aei = model.materialize_actives(b)
    assert torch.equal(ani.cpu(), ob.active_node_indices)
    assert torch.equal(aei.cpu(), ob.active_edge_indices)
    assert torch.equal(b.active_row.cpu(), ob.full_edge_index[0][ob.active_edge_indices])
    assert torch.equal(b.active_col.cpu(), ob.full_edge_index[1][ob.active_edge_indices])
    mask = torch.zeros(Nn, dtype=torch.uint8)
    mask[ob.active_node_indices] = 1
    assert torch.equal(b._state['flags'][:Nn].cpu(), mask)


def test_active_select_rng_matches_host_philox():
    """Counter-based uniforms: device draw == numpy Philox fed to the oracle (and shard-invariant)."""
    builder, T, wseed, useed = cases.CASES['ragged']
    graphs = builder()
    ob = orc.make_batch(graphs)
    s = orc.make_schedule(T)
    ob.log_full_edge_attr_t = orc.index_to_log_onehot(torch.zeros_like(ob.full_edge_attr))     # empty state
    from edgehip import capi
    model = H.build_model(orc.random_tgnn_params(wseed), T, cases.max_degree_of(graphs))
    for gid_base in (0, 5):
        b = H.graph_batch(graphs, with_edges=False, gid_base=gid_base)
        t, seed = 17, 0x1234567890abcdef
        model._p_sample_and_set_actives(b, t, None, capi.Rng(seed))
        u = H.rng_node_uniforms(seed, [g['n'] for g in graphs], t, gid_base)
        orc.p_sample_and_set_actives(s, ob, torch.full((ob.num_nodes,), t), orc.UniformTape([u]))
        ani, aei = model.materialize_actives(b)
        assert torch.equal(ani.cpu(), ob.active_node_indices)
        assert torch.equal(aei.cpu(), ob.active_edge_indices)
        assert len(ani) > 0


@pytest.mark.parametrize('case,tl', [('toy', [6, 4]), ('toy', [3, 0]), ('ragged', [0, 7, 47, 12, 1, 20]),
                                     ('polblogs1', [100]), ('polblogs1', [0]), ('polblogs1', [511])])
def test_forward_noising_and_train_actives(case, tl):
    """a4-a7 + a11: x_{t-1}, x_t, both degree vectors and the training active sets."""
    from edgehip import capi
    builder, T, wseed, useed = cases.CASES[case]
    graphs = builder()
    rng = np.random.default_rng(useed + 31)
    ob = orc.make_batch(graphs, torch.int64)
    s = orc.make_schedule(T)
    Nn, P = ob.num_nodes, ob.full_edge_attr.shape[0]
    us = [cases.uniforms(rng, Nn, 2), cases.uniforms(rng, P, 2), cases.uniforms(rng, Nn, 2), cases.uniforms(rng, P, 2)]
    t = torch.tensor(tl)
    orc.q_sample_and_set_xtmin1_xt_given_x0(s, ob, t.repeat_interleave(ob.nodes_per_graph),
                                            t.repeat_interleave(ob.edges_per_graph), orc.UniformTape(us))
    orc.q_set_actives(ob)
    model = H.build_model(orc.random_tgnn_params(wseed), T, cases.max_degree_of(graphs))
    b = H.graph_batch(graphs)
    tabs = model._tables(b.device, 0)
    a1, a2 = b.new_adj(), b.new_adj()
    tg = t.to(torch.int32).cuda()
    u1d, u3d = us[1].cuda(), us[3].cuda()      # keep the device copies alive across the call
    capi.check(capi.lib.edgehip_q_sample_edges(byref(b.layout), b.adj.data_ptr(), tg.data_ptr(),
                                               tabs['q_cum'].data_ptr(), tabs['q_one'].data_ptr(),
                                               u1d.data_ptr(), u3d.data_ptr(), None,
                                               tabs['lut'].data_ptr(), a1.data_ptr(), a2.data_ptr(), capi.stream_ptr()))
    assert torch.equal(b.pair_bits(a1).cpu(), ob.log_full_edge_attr_tmin1.argmax(1))
    assert torch.equal(b.pair_bits(a2).cpu(), ob.log_full_edge_attr_t.argmax(1))
    d1 = torch.empty(Nn, dtype=torch.int32, device='cuda'); d2 = torch.empty_like(d1)
    capi.lib.edgehip_degree_from_adj(byref(b.layout), a1.data_ptr(), d1.data_ptr(), capi.stream_ptr())
    capi.lib.edgehip_degree_from_adj(byref(b.layout), a2.data_ptr(), d2.data_ptr(), capi.stream_ptr())
    assert torch.equal(d1.cpu().long(), ob.degree_tmin1) and torch.equal(d2.cpu().long(), ob.degree_t)
    from edgehip.model import _state
    st = _state(b)
    capi.check(capi.lib.edgehip_active_select_train(byref(b.layout), d1.data_ptr(), d2.data_ptr(), st['flags'].data_ptr(),
                                                    st['act_nodes'].data_ptr(), st['act_count'].data_ptr(),
                                                    capi.stream_ptr()))
    model._set_offsets(b)
    ani, aei = model.materialize_actives(b)
    assert torch.equal(ani.cpu(), ob.active_node_indices)
    assert torch.equal(aei.cpu(), ob.active_edge_indices)


def test_edge_resample_given_logprobs_is_bit_exact():
    """a16: 'sampled edges bit-exact given logits + uniforms' on 14k active pairs (golden: reference run)."""
    g = H.golden('step_polblogs1_t100.npz')
    builder, T, wseed, useed = cases.CASES['polblogs1']
    graphs = builder()
    rng = np.random.default_rng(useed)
    Nn = graphs[0]['n']
    P = Nn * (Nn - 1) // 2
    cases.uniforms(rng, Nn, 2), cases.uniforms(rng, P, 2)
    u_act, u_node = cases.uniforms(rng, Nn, 2), cases.uniforms(rng, Nn, 2)
    A = len(g['active_edge_indices'])
    u_pair = cases.uniforms(rng, A, 2)
    model = H.build_model(orc.random_tgnn_params(wseed), T, int(g['max_degree']))
    b = H.graph_batch(graphs)
    H.set_state_from_pair_indices(b, g['xt_before'])
    want = {}
    # the reference's logits go in; log_softmax + Gumbel + overwrite happen on the device
    model.p_sample(b, 100, u_act.cuda(), u_pair.cuda(), None, torch.from_numpy(g['logits']).cuda(), want)
    assert np.array_equal(want['bits'].cpu().numpy().astype(np.int64), g['sampled_bits'])
    assert np.array_equal(b.pair_bits().nonzero(as_tuple=True)[0].cpu().numpy(), g['xt_after'])
