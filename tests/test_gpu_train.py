"""GPU: the training ops (csrc/train.hip) against plain PyTorch fp32/fp64 references of the same op, and the
gradient of the whole loss against the oracle's autograd (torch CPU restatement of the reference).

Tolerance: gradients are sums of fp32 products in a different order than torch's, so they are compared
as max|gpu - ref| / max|ref| per tensor with GRAD_TOL = 2e-4 (the forward bound is 1e-4)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import cases
import helpers as H
from oracle import edge_oracle as orc

pytestmark = pytest.mark.gpu
byref = C.byref
GRAD_TOL = 2e-4


def _rand_graphs(seed, sizes, p=0.15):
    rng = np.random.default_rng(seed)
    gs = []
    for n in sizes:
        iu = np.triu_indices(n, 1)
        keep = rng.random(len(iu[0])) < p
        gs.append(dict(n=n, edges=np.stack([iu[0][keep], iu[1][keep]], 1).astype(np.int64)))
    return gs


def _dense_adj(graphs):
    Nn = sum(g['n'] for g in graphs)
    A = torch.zeros(Nn, Nn, dtype=torch.bool)
    o = 0
    for g in graphs:
        e = torch.as_tensor(g['edges'])
        if len(e):
            A[o + e[:, 0], o + e[:, 1]] = True
            A[o + e[:, 1], o + e[:, 0]] = True
        o += g['n']
    return A


@pytest.mark.parametrize('M,K,N', [(1, 64, 64), (37, 128, 512), (1222, 192, 64), (4888, 64, 192), (300, 256, 128), (5, 512, 64),
                                   (8192, 64, 64), (39104, 128, 512), (20011, 256, 192)])      # large M: the 64 x 64 dW kernel
def test_linear_backward_vs_torch(M, K, N):
    from edgehip import train
    g = torch.Generator().manual_seed(M + K + N)
    x = torch.randn(M, K, generator=g).cuda().requires_grad_()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda().requires_grad_()
    b = torch.randn(N, generator=g).cuda().requires_grad_()
    dy = torch.randn(M, N, generator=g).cuda()
    y = train._Linear.apply(x, w, b)
    y.backward(dy)
    xr, wr, br = (t.detach().double().cpu().requires_grad_() for t in (x, w, b))
    F.linear(xr, wr, br).backward(dy.double().cpu())
    assert H.rel_err(x.grad, xr.grad) < 2e-6
    assert H.rel_err(w.grad, wr.grad) < 2e-6
    assert H.rel_err(b.grad, br.grad) < 2e-6


@pytest.mark.parametrize('kind', [1, 2])
def test_activation_vs_torch(kind):
    from edgehip import train
    x = (torch.randn(1000, 64, generator=torch.Generator().manual_seed(kind)) * 4).cuda().requires_grad_()
    dy = torch.randn(1000, 64, generator=torch.Generator().manual_seed(5)).cuda()
    y = train.act(x, kind)
    y.backward(dy)
    xr = x.detach().double().cpu().requires_grad_()
    yr = F.silu(xr) if kind == 1 else xr * torch.tanh(F.softplus(xr))
    yr.backward(dy.double().cpu())
    assert H.rel_err(y, yr) < 1e-6
    assert H.rel_err(x.grad, xr.grad) < 1e-6


@pytest.mark.parametrize('heads', [1, 2, 4, 8])
def test_attention_fwd_bwd_vs_dense_torch(heads):
    """TransformerConv semantics on a ragged batch with isolated nodes."""
    from edgehip import train
    graphs = _rand_graphs(heads, [1, 33, 70, 5], p=0.12)
    graphs[2]['edges'] = graphs[2]['edges'][graphs[2]['edges'].min(1) > 3]        # nodes 0..3 isolated
    b = H.graph_batch(graphs)
    Nn = b.num_nodes
    g = torch.Generator().manual_seed(11)
    q, k, v = (torch.randn(Nn, heads * 64, generator=g) for _ in range(3))
    skip, dout = torch.randn(Nn, 64, generator=g), torch.randn(Nn, 64, generator=g)
    qd, kd, vd, sd = (t.cuda().requires_grad_() for t in (q, k, v, skip))
    out = train._Attention.apply(qd, kd, vd, sd, b.layout, b.adj, heads)
    out.backward(dout.cuda())
    A = _dense_adj(graphs)
    qr, kr, vr, sr = (t.double().requires_grad_() for t in (q, k, v, skip))
    s = torch.einsum('ihc,jhc->hij', qr.view(Nn, heads, 64), kr.view(Nn, heads, 64)) / 8.0
    s = s.masked_fill(~A[None], float('-inf'))
    a = torch.softmax(s, dim=-1)
    a = torch.where(A[None].expand_as(a), a, torch.zeros_like(a))                 # rows with no neighbour -> 0
    ref = torch.einsum('hij,jhc->ihc', a, vr.view(Nn, heads, 64)).mean(1) + sr
    ref.backward(dout.double())
    assert H.rel_err(out, ref) < 2e-6
    for got, want in ((qd.grad, qr.grad), (kd.grad, kr.grad), (vd.grad, vr.grad), (sd.grad, sr.grad)):
        assert H.rel_err(got, want) < 5e-6


def test_gru_gate_vs_torch():
    from edgehip import train
    g = torch.Generator().manual_seed(3)
    gi, gh, h, dh = torch.randn(777, 192, generator=g), torch.randn(777, 192, generator=g), torch.randn(777, 64, generator=g), \
        torch.randn(777, 64, generator=g)
    a, b_, c = (t.cuda().requires_grad_() for t in (gi, gh, h))
    out = train._GruGate.apply(a, b_, c)
    out.backward(dh.cuda())
    ar, br, cr = (t.double().requires_grad_() for t in (gi, gh, h))
    i_r, i_z, i_n = ar.chunk(3, 1)
    h_r, h_z, h_n = br.chunk(3, 1)
    r, z = torch.sigmoid(i_r + h_r), torch.sigmoid(i_z + h_z)
    n = torch.tanh(i_n + r * h_n)
    ref = (1 - z) * n + z * cr
    ref.backward(dh.double())
    assert H.rel_err(out, ref) < 1e-6
    for got, want in ((a.grad, ar.grad), (b_.grad, br.grad), (c.grad, cr.grad)):
        assert H.rel_err(got, want) < 2e-6


def test_assemble_segment_pair_embed_vs_torch():
    from edgehip import train
    graphs = _rand_graphs(2, [40, 1, 77])
    b = H.graph_batch(graphs)
    L, Nn, B = b.layout, b.num_nodes, b.num_graphs
    bi = b.batch.long().cpu()
    g = torch.Generator().manual_seed(8)
    h, tr = torch.randn(Nn, 64, generator=g), torch.randn(Nn, 192, generator=g)
    cx, tg = torch.randn(B, 64, generator=g), torch.randn(B, 64, generator=g)
    # assemble, both tail kinds
    for tail in ('graph', 'rows'):
        hd, cd = h.cuda().requires_grad_(), cx.cuda().requires_grad_()
        td = (tg if tail == 'graph' else tr).cuda().requires_grad_()
        out = train._Assemble.apply(hd, cd, td if tail == 'graph' else None, td if tail == 'rows' else None, L)
        hr, cr = h.double().requires_grad_(), cx.double().requires_grad_()
        t_r = (tg if tail == 'graph' else tr).double().requires_grad_()
        ref = torch.cat([hr + cr[bi], t_r[bi] if tail == 'graph' else t_r], 1)
        dy = torch.randn(ref.shape, generator=g)
        out.backward(dy.cuda())
        ref.backward(dy.double())
        assert H.rel_err(out, ref) < 1e-6
        for got, want in ((hd.grad, hr.grad), (cd.grad, cr.grad), (td.grad, t_r.grad)):
            assert H.rel_err(got, want) < 2e-6
    # segment mean with addend
    xd, ad = h.cuda().requires_grad_(), cx.cuda().requires_grad_()
    out = train._SegmentMean.apply(xd, ad, L)
    xr, ar = h.double().requires_grad_(), cx.double().requires_grad_()
    cnt = torch.bincount(bi, minlength=B).double()
    ref = ar + torch.zeros(B, 64, dtype=torch.float64).index_add(0, bi, xr) / cnt[:, None]
    dy = torch.randn(B, 64, generator=g)
    out.backward(dy.cuda())
    ref.backward(dy.double())
    assert H.rel_err(out, ref) < 1e-6 and H.rel_err(xd.grad, xr.grad) < 1e-6 and H.rel_err(ad.grad, ar.grad) < 1e-6
    # pair gather-add
    A = 500
    row = torch.randint(0, Nn, (A,), generator=g)
    col = torch.randint(0, Nn, (A,), generator=g)
    z = torch.randn(Nn, 128, generator=g)
    zd = z.cuda().requires_grad_()
    e = train._PairGatherAdd.apply(zd, row.cuda(), col.cuda())
    zr = z.double().requires_grad_()
    er = zr[row] + zr[col]
    dy = torch.randn(A, 128, generator=g)
    e.backward(dy.cuda())
    er.backward(dy.double())
    assert H.rel_err(e, er) < 1e-6 and H.rel_err(zd.grad, zr.grad) < 2e-6
    # embeddings
    deg = torch.randint(0, 12, (Nn,), generator=g, dtype=torch.int32)
    flags = torch.randint(0, 2, (Nn,), generator=g, dtype=torch.uint8)
    dmax = 9
    ws = [torch.randn(64, 1, generator=g), torch.randn(64, generator=g), torch.randn(64, 1, generator=g),
          torch.randn(64, generator=g), torch.randn(2, 64, generator=g)]
    wd = [w.cuda().requires_grad_() for w in ws]
    X0 = train._Embed.apply(*wd, L, deg.cuda(), b.degree_i32, flags.cuda(), dmax)
    wr = [w.double().requires_grad_() for w in ws]
    st = (deg.clamp(max=dmax + 1).double() / dmax)[:, None]
    s0 = (b.degree_i32[:Nn].cpu().double() / dmax)[:, None]
    ref = torch.cat([F.linear(st, wr[0], wr[1]), F.linear(s0, wr[2], wr[3]), F.embedding(flags.long(), wr[4])], 1)
    dy = torch.randn(Nn, 192, generator=g)
    X0.backward(dy.cuda())
    ref.backward(dy.double())
    assert H.rel_err(X0, ref) < 1e-6
    for got, want in zip(wd, wr):
        assert H.rel_err(got.grad, want.grad) < 2e-6


def test_edge_ce_vs_torch():
    from edgehip import train
    g = torch.Generator().manual_seed(21)
    A, B = 3001, 3
    e2, W4, b4 = torch.randn(A, 64, generator=g), torch.randn(2, 64, generator=g) / 8, torch.randn(2, generator=g)
    tgt = torch.randint(0, 2, (A,), generator=g, dtype=torch.uint8)
    pg = torch.sort(torch.randint(0, B, (A,), generator=g, dtype=torch.int32))[0]
    gk = torch.randn(B, generator=g)
    ed, wd, bd = (t.cuda().requires_grad_() for t in (e2, W4, b4))
    ce, logits = train._EdgeCE.apply(ed, wd, bd, tgt.cuda(), pg.cuda(), B)
    ce.backward(gk.cuda())
    er, wr, br = (t.double().requires_grad_() for t in (e2, W4, b4))
    lr = F.linear(er, wr, br)
    cer = torch.zeros(B, dtype=torch.float64).index_add(0, pg.long(), F.cross_entropy(lr, tgt.long(), reduction='none'))
    cer.backward(gk.double())
    assert H.rel_err(logits, lr) < 1e-6 and H.rel_err(ce, cer) < 1e-5
    for got, want in ((ed.grad, er.grad), (wd.grad, wr.grad), (bd.grad, br.grad)):
        assert H.rel_err(got, want) < 5e-6


@pytest.mark.parametrize('name,case', [('loss_toy', 'toy'), ('loss_toy_mid', 'toy'), ('loss_ragged', 'ragged'),
                                       ('loss_polblogs1', 'polblogs1')])
@pytest.mark.parametrize('form', ['fused', 'ops'])
def test_loss_gradient_vs_oracle_autograd(name, case, form):
    """elbo_bpd(model, batch).backward() against torch autograd through the oracle on the same t, pt and
    uniforms (the golden fixture pins the forward values of this very configuration).  form 'fused': the
    network is one autograd node (edgehip_train_forward / _backward); 'ops': one node per op."""
    import edgehip
    g = H.golden(name + '.npz')
    builder, T, wseed, useed = cases.CASES[case]
    graphs = builder()
    rng = np.random.default_rng(useed + 1)
    Nn = sum(x['n'] for x in graphs)
    P = sum(x['n'] * (x['n'] - 1) // 2 for x in graphs)
    us = [cases.uniforms(rng, Nn, 2), cases.uniforms(rng, P, 2), cases.uniforms(rng, Nn, 2), cases.uniforms(rng, P, 2)]
    t, pt = torch.from_numpy(g['t']), torch.from_numpy(g['pt'])
    params = orc.random_tgnn_params(wseed)
    # oracle
    po = {k: v.clone().requires_grad_() for k, v in params.items()}
    o = orc.EdgeOracle(po, T, int(g['max_degree']))
    ob = orc.make_batch(graphs, degree_dtype=torch.int64)
    ob.log_node_attr = orc.index_to_log_onehot(ob.node_attr)
    ob.log_full_edge_attr = orc.index_to_log_onehot(ob.full_edge_attr)
    nv_ref = o.loss(ob, t, pt, orc.UniformTape(us), training=True)
    orc.elbo_bpd(nv_ref, ob.num_entries).backward()
    # product
    model = H.build_model(params, T, int(g['max_degree']))
    model.train()
    model.train_form = form
    b = H.graph_batch(graphs, degree_dtype=np.int64)
    want = {}
    nv = model.log_prob(b, t, pt, us[1].cuda(), us[3].cuda(), want)
    assert nv.requires_grad
    np.testing.assert_allclose(nv.detach().cpu().numpy(), g['neg_vb'], rtol=1e-4, atol=1e-12)
    if len(g['logits']):
        H.check_logits(want['logits'], g['logits'], 1e-4)
    bpd = -nv.sum() / (np.log(2) * b.num_entries)
    bpd.backward()
    named = dict(model._denoise_fn.named_parameters())
    checked = 0
    gmax = max(float(r.grad.abs().max()) for r in po.values() if r.grad is not None)
    for k, ref in po.items():
        if ref.grad is None or float(ref.grad.abs().max()) == 0.0:
            assert named[k].grad is None or float(named[k].grad.abs().max()) == 0.0, k
            continue
        assert named[k].grad is not None, k
        # gradients that are zero in exact arithmetic (the key bias: softmax is shift-invariant) are pure
        # rounding noise on both sides; measure those against the global gradient scale
        scale = max(float(ref.grad.abs().max()), 1e-6 * gmax)
        err = float((named[k].grad.detach().cpu().double() - ref.grad.double()).abs().max()) / scale
        assert err < GRAD_TOL, (k, err)
        checked += 1
    assert checked >= (80 if len(g['logits']) else 0)
    # the same loss through the fused inference kernels
    with torch.no_grad():
        nv2 = model.log_prob(H.graph_batch(graphs, degree_dtype=np.int64), t, pt, us[1].cuda(), us[3].cuda())
    np.testing.assert_allclose(nv2.cpu().numpy(), nv.detach().cpu().numpy(), rtol=1e-4, atol=1e-12)


@pytest.mark.parametrize('arena', [True, False])
def test_train_step_matches_oracle_adam_and_learns(arena):
    """arena: gradients in one flat buffer the fused backward adds into (train.GradArena).
    One optimisation step (experiment.py:29-37 of the reference): parameters after Adam equal the ones
    obtained from the oracle's autograd gradients; a few more steps on the same (t, uniforms) lower the loss."""
    import edgehip
    g = H.golden('loss_ragged.npz')
    builder, T, wseed, useed = cases.CASES['ragged']
    graphs = builder()
    rng = np.random.default_rng(useed + 1)
    Nn = sum(x['n'] for x in graphs)
    P = sum(x['n'] * (x['n'] - 1) // 2 for x in graphs)
    us = [cases.uniforms(rng, Nn, 2), cases.uniforms(rng, P, 2), cases.uniforms(rng, Nn, 2), cases.uniforms(rng, P, 2)]
    t, pt = torch.from_numpy(g['t']), torch.from_numpy(g['pt'])
    params = orc.random_tgnn_params(wseed)
    po = {k: v.clone().requires_grad_() for k, v in params.items()}
    o = orc.EdgeOracle(po, T, int(g['max_degree']))
    ob = orc.make_batch(graphs, degree_dtype=torch.int64)
    ob.log_node_attr = orc.index_to_log_onehot(ob.node_attr)
    ob.log_full_edge_attr = orc.index_to_log_onehot(ob.full_edge_attr)
    ref_opt = torch.optim.Adam(list(po.values()), lr=1e-3, betas=(0.9, 0.999))
    orc.elbo_bpd(o.loss(ob, t, pt, orc.UniformTape(us), training=True), ob.num_entries).backward()
    torch.nn.utils.clip_grad_value_([p for p in po.values() if p.grad is not None], 1.0)
    ref_opt.step()

    model = H.build_model(params, T, int(g['max_degree']))
    opt = edgehip.make_optimizer(model)
    b = H.graph_batch(graphs, degree_dtype=np.int64)
    kw = dict(t=t, pt=pt, u_tm1=us[1].cuda(), u_t=us[3].cuda())
    if arena:
        kw['state'] = {}
    losses = [float(edgehip.train_step(model, b, opt, clip_value=1.0, **kw))]
    # Adam's first step moves every weight by lr * g / (|g| + 1e-8): where |g| ~ 1e-8 the step amplifies the
    # rounding noise of g, so the bound is 5% of lr element-wise and 0.1% of lr on average
    gmax = max(float(r.grad.abs().max()) for r in po.values() if r.grad is not None)
    for k, v in model._denoise_fn.named_parameters():
        d = (v.detach().cpu() - po[k].detach()).abs()
        if po[k].grad is None or float(po[k].grad.abs().max()) < 1e-6 * gmax:
            # a gradient that is zero in exact arithmetic (the key bias: softmax is shift-invariant) is rounding noise
            # on both sides, and Adam turns noise of any size into a step of up to lr: only that bound holds
            assert float(d.max()) <= 1.1e-3, (k, float(d.max()))
            continue
        assert float(d.max()) < 5e-5 and float(d.mean()) < 1e-6, (k, float(d.max()), float(d.mean()))
    for _ in range(5):
        losses.append(float(edgehip.train_step(model, b, opt, clip_value=1.0, **kw)))
    assert losses[-1] < losses[0]
    # the sampler refolds after the parameters moved (version counters) and still runs
    model.eval()
    with torch.no_grad():
        nv = model.log_prob(H.graph_batch(graphs, degree_dtype=np.int64), t, pt, us[1].cuda(), us[3].cuda())
    assert abs(float(-nv.sum() / (np.log(2) * P)) - losses[-1]) < 0.2 * abs(losses[0] - losses[-1]) + 1e-3
    # ... with the weights of NOW: torch's fused Adam leaves the version counters alone, the fold cache must not
    fresh = H.build_model(params, T, int(g['max_degree']))
    fresh.load_state_dict(model.state_dict())
    fresh.eval()
    with torch.no_grad():
        nv3 = fresh.log_prob(H.graph_batch(graphs, degree_dtype=np.int64), t, pt, us[1].cuda(), us[3].cuda())
    assert torch.allclose(nv, nv3, rtol=1e-5, atol=1e-9)


def test_fused_and_op_forms_agree_and_two_live_forwards():
    """The one-node form (edgehip_train_forward/_backward) against the node-per-op form on a ragged batch with
    shared (t, uniforms): same loss, same gradients to float-atomic noise; a second forward on the same batch
    before the first backward gets a workspace of its own (both backwards still correct)."""
    graphs = cases.ragged_graphs(7) + cases.ragged_graphs(8)
    T = 48
    dmax = cases.max_degree_of(graphs)
    params = orc.random_tgnn_params(11)
    rng = np.random.default_rng(5)
    P = sum(x['n'] * (x['n'] - 1) // 2 for x in graphs)
    u1, u3 = cases.uniforms(rng, P, 2).cuda(), cases.uniforms(rng, P, 2).cuda()
    t = torch.tensor([5, 9, 3, 20, 1, 13] * 2)
    pt = torch.full((len(graphs),), 1.0 / T)
    grads = {}
    for form in ('ops', 'fused'):
        model = H.build_model(params, T, dmax)
        model.train()
        model.train_form = form
        b = H.graph_batch(graphs, degree_dtype=np.int64)
        nv = model.log_prob(b, t, pt, u1, u3)
        if form == 'fused':
            nv_b = model.log_prob(b, t, pt, u1, u3)           # second live forward on the same batch object
            assert torch.allclose(nv, nv_b, rtol=1e-6, atol=1e-9)
        (-nv.sum() / (np.log(2) * b.num_entries)).backward()
        grads[form] = {k: v.grad.detach().clone() for k, v in model._denoise_fn.named_parameters() if v.grad is not None}
        grads[form + '_nv'] = nv.detach().clone()
        if form == 'fused':
            for v in model._denoise_fn.parameters():
                v.grad = None
            (-nv_b.sum() / (np.log(2) * b.num_entries)).backward()
            grads['fused_b'] = {k: v.grad.detach().clone() for k, v in model._denoise_fn.named_parameters() if v.grad is not None}
    assert torch.allclose(grads['ops_nv'], grads['fused_nv'], rtol=1e-5, atol=1e-9)
    gmax = max(float(g.abs().max()) for g in grads['ops'].values())
    assert gmax > 0
    for k, g in grads['ops'].items():
        scale = max(float(g.abs().max()), 1e-6 * gmax)
        for other in ('fused', 'fused_b'):
            err = float((grads[other][k] - g).abs().max()) / scale
            assert err < 2e-4, (k, other, err)


def test_second_live_forward_with_other_t_does_not_corrupt_the_first_backward():
    """Two differentiable losses alive on ONE batch object, the second with different t and uniforms: the first
    backward must still see its own degrees / active flags (they feed the embedding gradients).  Reference = the
    same first loss evaluated alone on a fresh batch."""
    graphs = cases.ragged_graphs(7) + cases.ragged_graphs(8)
    T = 48
    dmax = cases.max_degree_of(graphs)
    params = orc.random_tgnn_params(11)
    rng = np.random.default_rng(6)
    P = sum(x['n'] * (x['n'] - 1) // 2 for x in graphs)
    ua = [cases.uniforms(rng, P, 2).cuda() for _ in range(2)]
    ub = [cases.uniforms(rng, P, 2).cuda() for _ in range(2)]
    ta = torch.tensor([5, 9, 3, 20, 1, 13] * 2)
    tb = torch.tensor([30, 2, 17, 4, 40, 25] * 2)
    pt = torch.full((len(graphs),), 1.0 / T)

    def grads_of(model):
        return {k: v.grad.detach().clone() for k, v in model._denoise_fn.named_parameters() if v.grad is not None}
    for form in ('fused', 'ops'):
        alone = H.build_model(params, T, dmax)
        alone.train()
        alone.train_form = form
        b0 = H.graph_batch(graphs, degree_dtype=np.int64)
        nv0 = alone.log_prob(b0, ta, pt, ua[0], ua[1])
        (-nv0.sum() / (np.log(2) * b0.num_entries)).backward()
        ref = grads_of(alone)
        model = H.build_model(params, T, dmax)
        model.train()
        model.train_form = form
        b = H.graph_batch(graphs, degree_dtype=np.int64)
        nv1 = model.log_prob(b, ta, pt, ua[0], ua[1])
        nv2 = model.log_prob(b, tb, pt, ub[0], ub[1])          # overwrote deg_t / flags of the first before the fix
        assert not torch.allclose(nv1, nv2)
        (-nv1.sum() / (np.log(2) * b.num_entries)).backward()
        got = grads_of(model)
        gmax = max(float(g.abs().max()) for g in ref.values())
        for k in ('embedding_t.weight', 'embedding_0.weight', 'embedding_sel.weight', 'node_in.0.weight'):
            scale = max(float(ref[k].abs().max()), 1e-6 * gmax)
            err = float((got[k] - ref[k]).abs().max()) / scale
            assert err < 2e-4, (form, k, err)
        del nv2


def test_train_entry_points_reject_bad_arguments():
    from edgehip import capi
    graphs = cases.ragged_graphs(3)
    model = H.build_model(orc.random_tgnn_params(5), 48, cases.max_degree_of(graphs))
    b = H.graph_batch(graphs, degree_dtype=np.int64)
    Pm = model._denoise_fn.raw_params()
    need = capi.lib.edgehip_train_workspace_bytes(C.byref(b.layout), C.byref(Pm), 10)
    assert need > 0 and capi.lib.edgehip_train_workspace_bytes(C.byref(b.layout), C.byref(Pm), -1) == 0
    z = torch.zeros(64, device='cuda')
    zi = torch.zeros(b.num_nodes + 32, dtype=torch.int32, device='cuda')
    zb = torch.zeros(b.num_nodes + 32, dtype=torch.uint8, device='cuda')
    tg = torch.zeros(b.num_graphs, dtype=torch.int32, device='cuda')
    ws = torch.empty(1024, dtype=torch.uint8, device='cuda')
    rc = capi.lib.edgehip_train_forward(C.byref(b.layout), C.byref(Pm), b.adj.data_ptr(), zi.data_ptr(), b.degree_i32.data_ptr(),
                                        zb.data_ptr(), tg.data_ptr(), None, None, 0, None, None, z.data_ptr(), None,
                                        ws.data_ptr(), ws.numel(), capi.stream_ptr())
    assert rc == capi.ERR_WORKSPACE
    rc = capi.lib.edgehip_train_forward(C.byref(b.layout), C.byref(Pm), None, zi.data_ptr(), b.degree_i32.data_ptr(),
                                        zb.data_ptr(), tg.data_ptr(), None, None, 0, None, None, z.data_ptr(), None,
                                        ws.data_ptr(), ws.numel(), capi.stream_ptr())
    assert rc == capi.ERR_ARG
    torch.cuda.synchronize()


def test_prelaunched_node_part_hit_and_miss_give_the_same_gradients():
    """Steps on ONE batch object with different t: the node part of the forward is queued ahead of the pair-count
    readback from the second step on (capacity of the earlier steps); a step with more active pairs than that
    capacity discards the queued part and redoes it.  Every step's loss and gradients equal those of a fresh batch
    (no prelaunch) on the same (t, uniforms)."""
    graphs = cases.ragged_graphs(7) + cases.ragged_graphs(9)
    T = 48
    dmax = cases.max_degree_of(graphs)
    params = orc.random_tgnn_params(13)
    rng = np.random.default_rng(8)
    P = sum(x['n'] * (x['n'] - 1) // 2 for x in graphs)
    u1, u3 = cases.uniforms(rng, P, 2).cuda(), cases.uniforms(rng, P, 2).cuda()
    pt = torch.full((len(graphs),), 1.0 / T)
    model = H.build_model(params, T, dmax)
    model.train()
    b = H.graph_batch(graphs, degree_dtype=np.int64)
    seen = []
    for t_val in (40, 40, 3, 20, 1):          # few active pairs first, then many (capacity miss), then fewer again
        t = torch.full((len(graphs),), t_val)
        for v in model._denoise_fn.parameters():
            v.grad = None
        if t_val == 20:
            b._train_cap = 16                 # these graphs never outgrow a real capacity: force the miss
        want = {}
        nv = model.log_prob(b, t, pt, u1, u3, want)
        seen.append((int(want['logits'].shape[0]), getattr(b, '_train_cap', 0)))
        (-nv.sum() / (np.log(2) * b.num_entries)).backward()
        got = {k: v.grad.detach().clone() for k, v in model._denoise_fn.named_parameters() if v.grad is not None}
        ref_model = H.build_model(params, T, dmax)
        ref_model.train()
        fb = H.graph_batch(graphs, degree_dtype=np.int64)
        nv2 = ref_model.log_prob(fb, t, pt, u1, u3)
        (-nv2.sum() / (np.log(2) * fb.num_entries)).backward()
        assert torch.allclose(nv.detach(), nv2.detach(), rtol=1e-5, atol=1e-9), t_val
        ref = {k: v.grad for k, v in ref_model._denoise_fn.named_parameters() if v.grad is not None}
        if not ref:
            assert not got or all(float(g.abs().max()) == 0.0 for g in got.values())
            continue
        gmax = max(float(g.abs().max()) for g in ref.values())
        for k, g in ref.items():
            scale = max(float(g.abs().max()), 1e-6 * gmax)
            if scale == 0.0:                  # no active pair at this t: every gradient is exactly zero
                assert float(got[k].abs().max()) == 0.0, (t_val, k)
                continue
            assert float((got[k] - g).abs().max()) / scale < 2e-4, (t_val, k)
    net = model._denoise_fn
    assert net.pos_emb('cuda') is net.pos_emb(b.device) is net.pos_emb(torch.device('cuda', torch.cuda.current_device()))
    assert net.raw_params().pos_emb == net._pos_emb.data_ptr()      # 'cuda' vs 'cuda:0' once rebuilt (and freed) it per step
    assert max(a for a, _ in seen) > min(a for a, _ in seen)
    assert seen[3][0] > 16 and seen[3][1] > seen[3][0]         # the forced miss re-grew the capacity


def test_importance_time_sampling_switches_on_like_the_reference():
    """sample_time_method='importance' through train_step with no t supplied (diffusion_binomial_vanilla.py:248-269):
    uniform t with pt = 1/T until every Lt_count exceeds 10, then t ~ multinomial(sqrt(Lt_history + 1e-10) + 1e-4 with
    entry 0 := entry 1) with pt the probability of the drawn t.  The host-side shortcut of `_importance_ready` (no
    device sync per step) must agree with the direct `(Lt_count > 10).all()` at every step."""
    import edgehip
    graphs = cases.ragged_graphs(7) + cases.ragged_graphs(9)
    T = 48
    model = H.build_model(orc.random_tgnn_params(5), T, cases.max_degree_of(graphs))
    model.sample_time_method = 'importance'
    model.rng_seed = 77
    torch.manual_seed(5)
    b = H.graph_batch(graphs, degree_dtype=np.int64)
    opt = edgehip.make_optimizer(model)
    state, switched = {}, None
    for step in range(400):
        direct = bool((model.Lt_count > 10).all())
        want = {}
        edgehip.train_step(model, b, opt, clip_value=1.0, state=state, want=want)
        pt, t = want['pt'].cpu(), want['t'].cpu()
        if not direct:
            assert torch.allclose(pt, torch.full_like(pt, 1.0 / T)), step
        else:
            if switched is None:
                switched = step
            assert getattr(model, '_lt_ready', False)
            assert float(pt.min()) > 0 and not torch.allclose(pt, torch.full_like(pt, 1.0 / T)), step
            if step - switched >= 5:
                break
    assert switched is not None and switched >= 11 * T // len(graphs) - 1
    assert int(model.Lt_count.sum()) == (step + 1) * len(graphs)
    # pt of the NEXT draw follows the formula on the buffers as they are now
    Ls = torch.sqrt(model.Lt_history + 1e-10) + 0.0001
    Ls[0] = Ls[1]
    t2, pt2 = model._sample_time(len(graphs), b.device, 'importance')
    assert torch.allclose(pt2, (Ls / Ls.sum())[t2])


def test_sampling_after_training_steps_uses_the_updated_weights():
    """sample -> three optimizer steps -> sample on ONE model object: the second generation equals, bit for bit, the
    generation of a fresh model that loaded the trained state_dict (same seed), and differs from the first one."""
    import edgehip
    graphs = cases.ragged_graphs(7) + cases.ragged_graphs(9)
    T = 48
    dmax = cases.max_degree_of(graphs)
    params = orc.random_tgnn_params(6)
    model = H.build_model(params, T, dmax)
    model.rng_seed = 3
    torch.manual_seed(2)

    def generate(m):
        sb = H.graph_batch(graphs, with_edges=False, gid_base=4)
        m.eval()
        m.sample_batch(sb, seed=21, parts=2)
        return sb.edge_index.clone(), sb._slice_dict['edge_index'].clone()

    e0, s0 = generate(model)
    b = H.graph_batch(graphs, degree_dtype=np.int64)
    opt = torch.optim.Adam(model.parameters(), lr=5e-2, fused=True)     # NOT make_optimizer: no post-step hook
    for _ in range(3):
        opt.zero_grad(set_to_none=True)
        model.train()
        nv = model.log_prob(b)
        (-nv.sum() / (np.log(2) * b.num_entries)).backward()
        with torch.no_grad():                 # an evaluation between backward and step folds the OLD weights
            model.log_prob(H.graph_batch(graphs, degree_dtype=np.int64))
        opt.step()
    e1, s1 = generate(model)
    fresh = H.build_model(params, T, dmax)
    fresh.load_state_dict(model.state_dict())
    e2, s2 = generate(fresh)
    assert torch.equal(e1, e2) and torch.equal(s1, s2)
    assert e0.shape != e1.shape or not torch.equal(e0, e1)
    # sampler blocks, the count read-back and the backward's weight-gradient stream all come from ONE small pool of
    # side streams (streams created independently end up sharing a hardware queue)
    from edgehip import capi
    pool = capi._SIDE_STREAMS[torch.cuda.current_device()]
    assert 1 <= len(pool) <= capi.MAX_SIDE_STREAMS
    assert all(rb[0] in pool for rb in getattr(model, '_rb', {}).values())
    assert torch.cuda.current_device() in capi._TRAIN_SIDE_SET
    sb = H.graph_batch(graphs, with_edges=False, gid_base=4)
    model.sample_batch(sb, seed=21, parts=6)             # more blocks than streams: blocks share streams in turn
    assert torch.equal(sb.edge_index, e1) and len(capi._SIDE_STREAMS[torch.cuda.current_device()]) <= capi.MAX_SIDE_STREAMS


def test_alternating_batches_share_one_model_and_arena():
    """Two batch objects of different size stepped alternately through one model and one gradient arena (each batch
    keeps its own workspace / capacity / queued node part): loss and arena gradients of every step equal those of a
    fresh model and batch."""
    from edgehip import train
    sets = [cases.ragged_graphs(7), cases.ragged_graphs(9) + cases.ragged_graphs(4)]
    T = 48
    dmax = max(cases.max_degree_of(g) for g in sets)
    params = orc.random_tgnn_params(17)
    model = H.build_model(params, T, dmax)
    model.train()
    arena = train.GradArena(list(model.parameters()))
    batches = [H.graph_batch(g, degree_dtype=np.int64) for g in sets]
    for step, (which, t_val) in enumerate([(0, 5), (1, 12), (0, 20), (1, 3), (1, 30), (0, 2)]):
        graphs, b = sets[which], batches[which]
        t = torch.full((len(graphs),), t_val)
        pt = torch.full((len(graphs),), 1.0 / T)
        model.rng_seed = 40 + step
        arena.zero_and_attach()
        b._train_ws_busy = False
        with train.use_arena(arena):
            nv = model.log_prob(b, t, pt)
            (-nv.sum() / (np.log(2) * b.num_entries)).backward()
        fresh = H.build_model(params, T, dmax)
        fresh.train()
        fresh.rng_seed = 40 + step
        fb = H.graph_batch(graphs, degree_dtype=np.int64)
        nv2 = fresh.log_prob(fb, t, pt)
        (-nv2.sum() / (np.log(2) * fb.num_entries)).backward()
        assert torch.allclose(nv.detach(), nv2.detach(), rtol=1e-5, atol=1e-9), step
        ref = dict(fresh._denoise_fn.named_parameters())
        gmax = max(float(v.grad.abs().max()) for v in ref.values() if v.grad is not None)
        assert gmax > 0
        for k, v in model._denoise_fn.named_parameters():
            g = ref[k].grad
            if g is None:
                assert float(v.grad.abs().max()) == 0.0, (step, k)
                continue
            scale = max(float(g.abs().max()), 1e-6 * gmax)
            assert float((v.grad - g).abs().max()) / scale < 2e-4, (step, k)


def test_sampling_and_training_follow_the_callers_stream():
    """Everything the library queues goes to the caller's current stream (or to side streams ordered behind it): run
    inside `torch.cuda.stream(s)` behind a long-running kernel that delays the inputs, a generation and a training
    gradient equal the default-stream results (a launch on the wrong stream would run ahead of its inputs)."""
    graphs = cases.ragged_graphs(7) + cases.ragged_graphs(9)
    T = 48
    dmax = cases.max_degree_of(graphs)
    params = orc.random_tgnn_params(19)
    rng = np.random.default_rng(4)
    P = sum(x['n'] * (x['n'] - 1) // 2 for x in graphs)
    u1h, u3h = cases.uniforms(rng, P, 2), cases.uniforms(rng, P, 2)
    t = torch.full((len(graphs),), 6)
    pt = torch.full((len(graphs),), 1.0 / T)

    def work(delay):
        if delay:
            a = torch.randn(4096, 4096, device='cuda')
            for _ in range(40):
                a = (a @ a).clamp_(-1, 1)          # ~tens of ms ahead of everything below on this stream
        model = H.build_model(params, T, dmax)
        sb = H.graph_batch(graphs, with_edges=False, gid_base=1)
        model.sample_batch(sb, seed=9, parts=3)
        model.train()
        b = H.graph_batch(graphs, degree_dtype=np.int64)
        grads = []
        for _ in range(2):                         # the second pass has its node part queued ahead of the pair count
            for v in model.parameters():
                v.grad = None
            nv = model.log_prob(b, t, pt, u1h.cuda(), u3h.cuda())
            (-nv.sum() / (np.log(2) * b.num_entries)).backward()
            grads.append(torch.cat([v.grad.reshape(-1) for v in model._denoise_fn.parameters() if v.grad is not None]))
        return sb.edge_index.clone(), nv.detach().clone(), grads

    e0, nv0, g0 = work(False)
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        e1, nv1, g1 = work(True)
    s.synchronize()
    assert torch.equal(e0, e1)
    assert torch.allclose(nv0, nv1, rtol=1e-5, atol=1e-9)
    scale = float(g0[0].abs().max())
    for a, b_ in zip(g0, g1):
        assert float((a - b_).abs().max()) / scale < 2e-4
    assert float((g0[0] - g0[1]).abs().max()) / scale < 2e-4


# ---- dropout (--dp_rate 0.1 of the reference's training templates, README.md:49,79) -----------------------------
@pytest.mark.parametrize('name,case', [('loss_ragged_dp', 'ragged'), ('loss_polblogs1_dp', 'polblogs1')])
@pytest.mark.parametrize('form', ['fused', 'ops'])
def test_dropout_loss_and_gradients_vs_reference_golden(name, case, form):
    """Training-mode dropout on both GRU-side activations (layers.py:357,361) with SUPPLIED keep-masks: the golden
    values come from the reference run with its nn.Dropout patched to the same masks; gradients against torch autograd
    through the oracle with the same masks."""
    g = H.golden(name + '.npz')
    builder, T, wseed, useed = cases.CASES[case]
    graphs = builder()
    rng = np.random.default_rng(useed + 1)
    Nn = sum(x['n'] for x in graphs)
    P = sum(x['n'] * (x['n'] - 1) // 2 for x in graphs)
    us = [cases.uniforms(rng, Nn, 2), cases.uniforms(rng, P, 2), cases.uniforms(rng, Nn, 2), cases.uniforms(rng, P, 2)]
    masks = cases.dropout_masks(useed, Nn, 0.1)
    t, pt = torch.from_numpy(g['t']), torch.from_numpy(g['pt'])
    params = orc.random_tgnn_params(wseed)
    po = {k: v.clone().requires_grad_() for k, v in params.items()}
    o = orc.EdgeOracle(po, T, int(g['max_degree']))
    o.dropout = orc.mask_dropout(masks, 0.1)
    ob = orc.make_batch(graphs, degree_dtype=torch.int64)
    ob.log_node_attr = orc.index_to_log_onehot(ob.node_attr)
    ob.log_full_edge_attr = orc.index_to_log_onehot(ob.full_edge_attr)
    orc.elbo_bpd(o.loss(ob, t, pt, orc.UniformTape(us), training=True), ob.num_entries).backward()

    import edgehip
    args = H.make_args(T, int(g['max_degree']))
    args.dp_rate = 0.1
    model = edgehip.get_model(args, None)
    model._denoise_fn.load_state_dict(params, strict=True)
    model = model.to('cuda').train()
    model.train_form = form
    b = H.graph_batch(graphs, degree_dtype=np.int64)
    want = {}
    dm = torch.stack(masks).cuda().contiguous()
    nv = model.log_prob(b, t, pt, us[1].cuda(), us[3].cuda(), want, dropout_masks=dm)
    np.testing.assert_allclose(nv.detach().cpu().numpy(), g['neg_vb'], rtol=1e-4, atol=1e-12)
    H.check_logits(want['logits'], g['logits'], 1e-4)
    # the dropout really acts: the dp 0 golden of the same case differs
    g0 = H.golden(name[:-3] + '.npz')
    assert H.rel_err(want['logits'], g0['logits']) > 1e-3
    (-nv.sum() / (np.log(2) * b.num_entries)).backward()
    named = dict(model._denoise_fn.named_parameters())
    gmax = max(float(r.grad.abs().max()) for r in po.values() if r.grad is not None)
    checked = 0
    for k, ref in po.items():
        if ref.grad is None or float(ref.grad.abs().max()) == 0.0:
            continue
        scale = max(float(ref.grad.abs().max()), 1e-6 * gmax)
        err = float((named[k].grad.detach().cpu().double() - ref.grad.double()).abs().max()) / scale
        assert err < GRAD_TOL, (k, err)
        checked += 1
    assert checked >= 80
    # eval mode: the dropout is off (torch.nn.Dropout follows the module mode) and the dp 0 golden comes back
    model.eval()
    with torch.no_grad():
        nv0 = model.log_prob(H.graph_batch(graphs, degree_dtype=np.int64), t, pt, us[1].cuda(), us[3].cuda())
    np.testing.assert_allclose(nv0.cpu().numpy(), g0['neg_vb'], rtol=1e-4, atol=1e-12)
    # train mode under no_grad keeps the dropout on, like the reference
    model.train()
    with torch.no_grad():
        nv1 = model.log_prob(H.graph_batch(graphs, degree_dtype=np.int64), t, pt, us[1].cuda(), us[3].cuda(), dropout_masks=dm)
    np.testing.assert_allclose(nv1.cpu().numpy(), g['neg_vb'], rtol=1e-4, atol=1e-12)


def test_generated_dropout_masks_are_bernoulli_scaled_reproducible_and_shard_invariant():
    """Without supplied masks the keep decisions come from Philox keyed (seed; graph global id, node, channel, slot)."""
    import edgehip
    from edgehip import capi
    n = [300, 500, 200]
    b = edgehip.GraphBatch(n, degree=[np.zeros(k) for k in n], device='cuda', gid_base=4)
    Nn = sum(n)
    x = torch.randn(Nn, 64, device='cuda')
    dp = capi.Dropout()
    dp.p, dp.seed = 0.1, 12345

    def run(batch, xin, slot, seed=12345):
        dp.seed = seed
        y = torch.empty_like(xin)
        capi.check(capi.lib.edgehip_dropout_fwd(C.byref(batch.layout), xin.data_ptr(), C.byref(dp), slot, y.data_ptr(),
                                                capi.stream_ptr()), 'dropout')
        return y
    y = run(b, x, 3)
    kept = y != 0
    assert abs(float(kept.float().mean()) - 0.9) < 0.01
    assert torch.equal(y[kept], (x * np.float32(1.0 / np.float32(0.9)))[kept])
    assert torch.equal(run(b, x, 3), y)                                   # reproducible
    assert not torch.equal(run(b, x, 4) != 0, kept)                       # another slot, another mask
    assert not torch.equal(run(b, x, 3, seed=7) != 0, kept)               # another seed
    # graph 1 alone, carrying its global id, draws the same mask
    b1 = edgehip.GraphBatch([500], degree=[np.zeros(500)], device='cuda', gid_base=5)
    assert torch.equal(run(b1, x[300:800].contiguous(), 3), y[300:800])
    # p = 0 is the identity; p outside [0, 1) is refused
    dp.p = 0.0
    assert torch.equal(run(b, x, 0), x)
    dp.p = 1.0
    assert capi.lib.edgehip_dropout_fwd(C.byref(b.layout), x.data_ptr(), C.byref(dp), 0, x.data_ptr(), None) == capi.ERR_ARG


def test_training_with_generated_dropout_runs_and_differs_between_steps():
    import edgehip
    graphs = cases.ragged_graphs(7)
    T, dmax = 48, cases.max_degree_of(graphs)
    args = H.make_args(T, dmax)
    args.dp_rate = 0.1
    model = edgehip.get_model(args, None)
    model._denoise_fn.load_state_dict(orc.random_tgnn_params(3), strict=True)
    model = model.to('cuda').train()
    b = H.graph_batch(graphs, degree_dtype=np.int64)
    P = b.num_pairs
    rng = np.random.default_rng(2)
    u1, u3 = cases.uniforms(rng, P, 2).cuda(), cases.uniforms(rng, P, 2).cuda()
    t = torch.tensor([5, 9, 3, 20, 1, 13])
    pt = torch.full((6,), 1.0 / T)
    model.rng_seed = 5
    a = model.log_prob(b, t, pt, u1, u3)
    c = model.log_prob(b, t, pt, u1, u3)              # next seed -> other masks
    model.rng_seed = 5
    a2 = model.log_prob(b, t, pt, u1, u3)
    # (per-graph CE sums use float atomics: equal to rounding, not bit for bit)
    assert torch.allclose(a, a2, rtol=1e-5, atol=0) and not torch.allclose(a, c, rtol=1e-4, atol=0)
    opt = edgehip.make_optimizer(model)
    l0 = float(edgehip.train_step(model, b, opt, clip_value=1.0, state={}))
    assert np.isfinite(l0)
