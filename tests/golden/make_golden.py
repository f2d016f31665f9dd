"""Generate golden vectors from the reference's own code (run in the build container only).

    python tests/golden/make_golden.py

Imports `/root/reference/diffusion/*.py` and `/root/reference/layers/layers.py`
unmodified, with three in-memory stand-in modules for third-party packages that
are not installed (`torch_scatter`, `torch_geometric`, `prettytable`).  The
stand-ins carry no EDGE logic: `scatter` is sum/mean by index, `degree` is a
count, and `TransformerConv` is the oracle's restatement of PyG's documented
operator (SURVEY.md App. B - its arithmetic is NOT pinned by the reference).
`torch.rand_like` is patched to replay supplied uniforms.

For every case the script (1) runs the reference, (2) runs `oracle/edge_oracle.py`
on the same inputs and asserts equality (bit-exact for the diffusion functions,
1e-6 for the GNN), (3) stores the reference's OUTPUTS under `tests/golden/*.npz`.
Inputs are regenerated from seeds by `cases.py`, so the fixtures stay small.
The reference never travels: tests read only the `.npz` files.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import cases  # noqa: E402
from oracle import edge_oracle as orc  # noqa: E402

REF = '/root/reference'


# ---------------------------------------------------------------- stand-ins
def _install_standins():
    ts = types.ModuleType('torch_scatter')

    def scatter(src, index, dim=-1, out=None, dim_size=None, reduce='sum'):
        assert out is None
        if dim_size is None:
            dim_size = int(index.max()) + 1
        if src.dim() > 1:
            assert dim == 0
        if reduce == 'sum':
            return orc.scatter_sum(src, index, dim_size)
        if reduce == 'mean':
            return orc.scatter_mean(src, index, dim_size)
        raise NotImplementedError(reduce)
    ts.scatter = scatter
    sys.modules['torch_scatter'] = ts

    pyg = types.ModuleType('torch_geometric')
    pyg.nn = types.ModuleType('torch_geometric.nn')
    pyg.nn.norm = types.ModuleType('torch_geometric.nn.norm')
    pyg.utils = types.ModuleType('torch_geometric.utils')

    class TransformerConv(torch.nn.Module):
        def __init__(self, in_channels, out_channels, heads=1, concat=True):
            super().__init__()
            assert not concat
            self.heads, self.out_channels = heads, out_channels
            self.lin_key = torch.nn.Linear(in_channels, heads * out_channels)
            self.lin_query = torch.nn.Linear(in_channels, heads * out_channels)
            self.lin_value = torch.nn.Linear(in_channels, heads * out_channels)
            self.lin_skip = torch.nn.Linear(in_channels, out_channels)

        def forward(self, x, edge_index):
            return orc.transformer_conv(x, edge_index, self.lin_query.weight, self.lin_query.bias,
                                        self.lin_key.weight, self.lin_key.bias, self.lin_value.weight,
                                        self.lin_value.bias, self.lin_skip.weight, self.lin_skip.bias,
                                        self.heads, self.out_channels)

    def degree(index, num_nodes=None, dtype=None):
        n = int(index.max()) + 1 if num_nodes is None else num_nodes
        return torch.zeros(n).index_add_(0, index, torch.ones(index.shape[0]))
    pyg.nn.TransformerConv = TransformerConv
    pyg.utils.degree = degree
    for m in (pyg, pyg.nn, pyg.nn.norm, pyg.utils):
        sys.modules[m.__name__] = m
    pt = types.ModuleType('prettytable')
    pt.PrettyTable = object
    sys.modules['prettytable'] = pt


_install_standins()
sys.path.insert(0, REF)
import diffusion.diffusion_base as ref_base  # noqa: E402
from diffusion.diffusion_binomial_active import BinomialDiffusionActive  # noqa: E402
from layers.layers import TGNN_degree_guided, SinusoidalPosEmb  # noqa: E402


class RefBatch(types.SimpleNamespace):
    def to(self, device):
        return self


def ref_batch(graphs, degree_dtype=torch.float32):
    b = orc.make_batch(graphs, degree_dtype)
    rb = RefBatch(**vars(b))
    rb._slice_dict, rb._inc_dict = {}, {'full_edge_index': None}
    return rb


class Replay:
    """Patched torch.rand_like: pops supplied uniforms in order."""

    def __init__(self, chunks):
        self.tape = orc.UniformTape(chunks)

    def __call__(self, x, *a, **k):
        return self.tape.take(x.shape).clone()


class SuppliedMaskDropout(torch.nn.Module):
    """Stands in for the reference network's `self.dropout = torch.nn.Dropout(p)` (layers.py:286): same place in the
    module tree, same calls (layers.py:357,361), same arithmetic y = x * keep / (1 - p) in training mode, identity in
    eval mode - but the Bernoulli keep-masks are SUPPLIED (call k of a forward takes masks[k]) so that the GPU
    path can be given the very same masks.  A forward pre-hook on the network rewinds the call counter."""

    def __init__(self, p, masks):
        super().__init__()
        self.p, self.masks, self.k = p, masks, 0

    def forward(self, x):
        if not self.training or self.p == 0:
            return x
        m = self.masks[self.k].to(x.dtype).reshape(x.shape)
        self.k += 1
        return x * (m * (1.0 / (1.0 - self.p)))


def build_ref_model(params, T, max_degree, schedule='linear', dropout=0., masks=None):
    net = TGNN_degree_guided(max_degree=max_degree, num_node_classes=2, num_edge_classes=2, dim=cases.DIM,
                             num_steps=T, num_heads=list(cases.HEADS), dropout=dropout, norm='None', gru=True,
                             degree=True, augmented_features={}, return_node_class=False)
    missing = net.load_state_dict(params, strict=True)
    if dropout > 0:
        assert isinstance(net.dropout, torch.nn.Dropout) and net.dropout.p == dropout
        net.dropout = SuppliedMaskDropout(dropout, masks)
        net.register_forward_pre_hook(lambda mod, inp: setattr(mod.dropout, 'k', 0))
    sched = {'linear': ref_base.linear_beta_schedule, 'cosine': ref_base.cosine_beta_schedule,
             'Tt1': ref_base.Tt1_beta_schedule}[schedule]
    m = BinomialDiffusionActive(2, 2, None, net, timesteps=T, loss_type='vb_ce_xt_prescribred_st',
                                parametrization='xt_prescribed_st', final_prob_node=[1 - 1e-12, 1e-12],
                                final_prob_edge=[1 - 1e-12, 1e-12], sample_time_method='importance',
                                noise_schedule=sched, device='cpu')
    m.eval()
    return m


def ones_idx(log_state):
    return log_state.argmax(1).nonzero(as_tuple=True)[0].numpy().astype(np.int64)


def eq(a, b, what):
    assert torch.equal(torch.as_tensor(a), torch.as_tensor(b)), f'oracle != reference: {what}'


def close(a, b, what, tol=1e-6):
    a, b = torch.as_tensor(a), torch.as_tensor(b)
    err = (a - b).abs().max().item() if a.numel() else 0.0
    ref = max(b.abs().max().item(), 1.0) if b.numel() else 1.0
    assert err <= tol * ref, f'oracle vs reference {what}: err {err} ref {ref}'


def save(name, **arrs):
    out = {k: (v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in arrs.items()}
    np.savez_compressed(os.path.join(HERE, name), **out)
    sz = os.path.getsize(os.path.join(HERE, name))
    print(f'  wrote {name} ({sz/1024:.1f} KiB)')


# ---------------------------------------------------------------- cases
def gen_schedule():
    print('schedule tables')
    out = {}
    for kind in ('linear', 'cosine', 'Tt1'):
        for T in (32, 64, 128, 512):
            m = build_ref_model(orc.random_tgnn_params(1), T, 10, kind)
            s = orc.make_schedule(T, kind)
            for nm in ('log_alpha', 'log_1_min_alpha', 'log_cumprod_alpha', 'log_1_min_cumprod_alpha'):
                eq(getattr(s, nm), getattr(m, nm), f'{kind} {T} {nm}')
                out[f'{kind}_{T}_{nm}'] = getattr(m, nm)
            eq(s.log_final_prob_edge, m.log_final_prob_edge, 'log_final_prob_edge')
            out['log_final_prob_edge'] = m.log_final_prob_edge
    # sinusoidal time embedding, all t (layers.py:45-60)
    for T in (32, 48, 512):
        pe = SinusoidalPosEmb(cases.DIM, T)(torch.arange(T))
        eq(orc.sinusoidal_pos_emb(torch.arange(T), cases.DIM, T), pe, 'pos emb')
        out[f'posemb_{T}'] = pe
    save('schedule.npz', **out)


def gen_posterior():
    print('active-node posterior grid')
    T = 512
    m = build_ref_model(orc.random_tgnn_params(1), T, 10)
    s = orc.make_schedule(T)
    ts = [0, 1, 2, 10, 100, 256, 400, 510, 511]
    nmax = 600
    tab = []
    for t in ts:
        d0 = torch.arange(nmax + 1, dtype=torch.float32)
        dt = torch.zeros(nmax + 1, dtype=torch.long)
        tn = torch.full((nmax + 1,), t, dtype=torch.long)
        r = m._q_posterior_actives(d0, dt, tn)
        eq(orc.q_posterior_actives(s, d0, dt, tn), r, f'posterior t={t}')
        tab.append(r)
    # int64 d0 (training batches, data_utils.py:62) gives the same numbers
    r64 = m._q_posterior_actives(torch.arange(5), torch.zeros(5, dtype=torch.long), torch.full((5,), 100))
    eq(r64, tab[4][:5], 'int64 d0')
    save('posterior.npz', t=np.array(ts), table=torch.stack(tab))


def noisy_state(model, rb, t, rng):
    """x_t ~ q(.|x0) with the reference's q_sample; returns uniforms used."""
    Nn, P = rb.num_nodes, rb.full_edge_attr.shape[0]
    un, ue = cases.uniforms(rng, Nn, 2), cases.uniforms(rng, P, 2)
    rb.log_node_attr = ref_base.index_to_log_onehot(rb.node_attr, 2)
    rb.log_full_edge_attr = ref_base.index_to_log_onehot(rb.full_edge_attr, 2)
    torch.rand_like = Replay([un, ue])
    ln, le = model.q_sample(rb, torch.full((Nn,), t), torch.full((P,), t))
    return ln, le


def gen_step(case, t_step):
    print(f'p_sample step: {case} t={t_step}')
    builder, T, wseed, useed = cases.CASES[case]
    graphs = builder()
    dmax = cases.max_degree_of(graphs)
    params = orc.random_tgnn_params(wseed)
    model = build_ref_model(params, T, dmax)
    rng = np.random.default_rng(useed)
    rb = ref_batch(graphs)
    # the state a sampler would hold: a noised version of the true graph, d0 = true degrees
    ln, le = noisy_state(model, rb, t_step, rng)
    rb.log_node_attr_t, rb.log_full_edge_attr_t = ln, le
    ob = orc.make_batch(graphs)
    ob.log_node_attr = rb.log_node_attr.clone()
    ob.log_node_attr_t, ob.log_full_edge_attr_t = ln.clone(), le.clone()
    xt_before = ones_idx(le)
    Nn = rb.num_nodes
    u_act = cases.uniforms(rng, Nn, 2)
    u_node = cases.uniforms(rng, Nn, 2)
    # run actives first to learn A, then draw the pair uniforms (same rng order in the tests)
    torch.rand_like = Replay([u_act])
    t_node = torch.full((Nn,), t_step)
    model._p_sample_and_set_actives(rb, t_node)
    A = rb.active_edge_indices.shape[0]
    u_pair = cases.uniforms(rng, A, 2)
    torch.rand_like = Replay([u_act, u_node, u_pair])
    with torch.no_grad():
        logits = model._denoise_fn(rb, t_node, None)[1] if A else torch.zeros(0, 2)
        out_node, out_edge = model.p_sample(rb, t_node, None)
    # oracle on the same inputs
    o = orc.EdgeOracle(params, T, dmax)
    rec = {}
    with torch.no_grad():
        on, oe = o.p_sample(ob, t_node, orc.UniformTape([u_act, u_node, u_pair]), rec)
    eq(rec['degree_t'], model._compute_degree(torch.zeros(le.shape[0], dtype=torch.long).index_fill_(
        0, torch.from_numpy(xt_before), 1), rb.full_edge_index, Nn), 'degree')
    eq(rec['active_node_indices'], rb.active_node_indices, 'active nodes')
    eq(rec['active_edge_indices'], rb.active_edge_indices, 'active pairs')
    if A:
        close(rec['logits'], logits, 'logits')
    eq(oe, out_edge, 'state after step')
    row = rb.full_edge_index[0][rb.active_edge_indices]
    col = rb.full_edge_index[1][rb.active_edge_indices]
    save(f'step_{case}_t{t_step}.npz', t=t_step, T=T, max_degree=dmax, xt_before=xt_before,
         degree_t=rec['degree_t'], active_node_indices=rb.active_node_indices,
         active_edge_indices=rb.active_edge_indices, active_row=row, active_col=col, logits=logits,
         sampled_bits=out_edge[rb.active_edge_indices].argmax(1), xt_after=ones_idx(out_edge))
    print(f'    E_t={len(xt_before)} K={len(rb.active_node_indices)} A={A} E_after={len(ones_idx(out_edge))}')


def gen_loss(case, t_list, tag='', dp=0.):
    print(f'train loss: {case} t={t_list} dp={dp}')
    builder, T, wseed, useed = cases.CASES[case]
    graphs = builder()
    dmax = cases.max_degree_of(graphs)
    params = orc.random_tgnn_params(wseed)
    masks = cases.dropout_masks(useed, sum(g['n'] for g in graphs), dp) if dp > 0 else None
    model = build_ref_model(params, T, dmax, dropout=dp, masks=masks)
    model.train()          # selects _train_loss (diffusion_base.py:183-187) and, with dp > 0, activates the dropout
    rng = np.random.default_rng(useed + 1)
    rb = ref_batch(graphs, torch.int64)       # training batches carry int64 degree (data_utils.py:62)
    ob = orc.make_batch(graphs, torch.int64)
    Nn, P = rb.num_nodes, rb.full_edge_attr.shape[0]
    t = torch.tensor(t_list)
    pt = torch.ones_like(t).float() / T
    us = [cases.uniforms(rng, Nn, 2), cases.uniforms(rng, P, 2), cases.uniforms(rng, Nn, 2), cases.uniforms(rng, P, 2)]
    model._sample_time = lambda b, device, method: (t, pt)
    torch.rand_like = Replay(us)
    with torch.no_grad():
        neg_vb = model.log_prob(rb)
        bpd = -neg_vb.sum() / (np.log(2) * rb.num_entries)   # diffusion/loss.py:12 (its module needs PyG to import)
        kl = model._compute_MC_KL_joint(rb, t, t.repeat_interleave(rb.nodes_per_graph), None)
        prior = model._kl_prior(rb)
        logits = model._denoise_fn(rb, t.repeat_interleave(rb.nodes_per_graph), None)[1]
    o = orc.EdgeOracle(params, T, dmax)
    if dp > 0:
        o.dropout = orc.mask_dropout(masks, dp)
    rec = {}
    with torch.no_grad():
        onv = o.loss(ob, t, pt, orc.UniformTape(us), training=True, record=rec)
    eq(ob.log_full_edge_attr_tmin1, rb.log_full_edge_attr_tmin1, 'x_{t-1}')
    eq(ob.log_full_edge_attr_t, rb.log_full_edge_attr_t, 'x_t')
    eq(ob.active_node_indices, rb.active_node_indices, 'active nodes')
    eq(ob.active_edge_indices, rb.active_edge_indices, 'active pairs')
    eq(ob.edge_predict_masks, rb.edge_predict_masks, 'edge_predict_masks')
    close(rec['logits'], logits, 'logits')
    close(rec['kl'], kl, 'kl', 1e-5)
    close(rec['kl_prior'], prior, 'kl_prior', 1e-6)
    close(onv, neg_vb, 'neg_vb', 1e-5)
    eq(o.Lt_count, model.Lt_count, 'Lt_count')
    close(o.Lt_history, model.Lt_history, 'Lt_history', 1e-5)
    save(f'loss_{case}{tag}.npz', t=t, pt=pt, T=T, max_degree=dmax, xtm1=ones_idx(rb.log_full_edge_attr_tmin1),
         xt=ones_idx(rb.log_full_edge_attr_t), degree_tmin1=rb.__dict__.get('degree_tmin1', ob.degree_tmin1),
         degree_t=ob.degree_t, active_node_indices=rb.active_node_indices,
         active_edge_indices=rb.active_edge_indices, logits=logits, kl=kl, kl_prior=prior, neg_vb=neg_vb,
         bpd=bpd, num_entries=rb.num_entries, Lt_history=model.Lt_history, Lt_count=model.Lt_count)
    print(f'    K={len(rb.active_node_indices)} A={len(rb.active_edge_indices)} neg_vb={neg_vb.tolist()}')


def gen_traj(case):
    print(f'sample trajectory: {case}')
    builder, T, wseed, useed = cases.CASES[case]
    graphs = builder()
    dmax = cases.max_degree_of(graphs)
    params = orc.random_tgnn_params(wseed)
    model = build_ref_model(params, T, dmax)
    rng = np.random.default_rng(useed + 2)
    empty = [dict(n=g['n'], edges=None, degree=synth_deg(g)) for g in graphs]
    rb = ref_batch(empty)
    Nn, P = rb.num_nodes, rb.full_edge_attr.shape[0]
    # uniforms: the tape depends on the trajectory (A per step), so a callback that draws
    # lazily from one rng in consumption order is used for the reference and logged for the tests.
    log = []

    def draw(x, *a, **k):
        u = cases.uniforms(rng, *x.shape)
        log.append(u)
        return u.clone()
    torch.rand_like = draw

    class Sampler:
        def sample(self, n):
            return rb
    model.initial_graph_sampler = Sampler()
    steps = {}
    orig = model.p_sample.__func__ if hasattr(model.p_sample, '__func__') else None

    def spy(batched_graph, t_node, t_edge):
        out = BinomialDiffusionActive.p_sample(model, batched_graph, t_node, t_edge)
        steps[int(t_node[0])] = dict(K=batched_graph.active_node_indices.clone(),
                                     A=batched_graph.active_edge_indices.clone(),
                                     state=ones_idx(out[1]))
        return out
    model.p_sample = spy
    import io
    import contextlib
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        res = model.sample(len(graphs))
    # oracle replays the logged uniforms
    ob = orc.make_batch(empty)
    o = orc.EdgeOracle(params, T, dmax)
    rec = {}
    with torch.no_grad():
        o.sample(ob, orc.UniformTape(log), rec)
    eq(ob.edge_index, res.edge_index, 'final edge_index')
    eq(ob.edge_slice, res._slice_dict['edge_index'], 'edge_slice')
    out = dict(T=T, max_degree=dmax, edge_index=res.edge_index, edge_slice=res._slice_dict['edge_index'],
               num_uniform_chunks=len(log))
    for t, s in steps.items():
        eq(rec[t]['active_node_indices'], s['K'], f'traj K t={t}')
        eq(rec[t]['active_edge_indices'], s['A'], f'traj A t={t}')
        out[f'K_{t}'] = s['K']
        out[f'A_{t}'] = s['A']
        out[f'state_{t}'] = s['state']
        if 'logits' in rec[t]:
            out[f'logits_{t}'] = rec[t]['logits']
    save(f'traj_{case}.npz', **out)
    print(f'    final edges per graph: {np.diff(res._slice_dict["edge_index"].numpy()).tolist()}')


def gen_traj_big(case='polblogs1'):
    """Whole T-step reference trajectory of ONE polblogs-scale graph (~10^6 Gumbel decisions on network logits),
    stored compactly: per step the active-node list, the pairs whose bit changed, the smallest decision margin
    with the reference logits of that pair, and full logits at a few steps.  Uniforms are NOT stored: the tests
    redraw them from the same seeded generator in the same consumption order."""
    print(f'polblogs-scale trajectory: {case}')
    builder, T, wseed, useed = cases.CASES[case]
    graphs = builder()
    dmax = cases.max_degree_of(graphs)
    params = orc.random_tgnn_params(wseed)
    model = build_ref_model(params, T, dmax)
    rng = np.random.default_rng(useed + 3)
    empty = [dict(n=g['n'], edges=None, degree=synth_deg(g)) for g in graphs]
    rb = ref_batch(empty)
    last = {}

    def draw(x, *a, **k):
        u = cases.uniforms(rng, *x.shape)
        last['u'] = u
        return u.clone()
    torch.rand_like = draw

    class Sampler:
        def sample(self, n):
            return rb
    model.initial_graph_sampler = Sampler()
    seen = {}
    model._denoise_fn.register_forward_hook(lambda mod, inp, out: seen.__setitem__('logits', out[1].detach().clone()))
    LOGIT_STEPS = (450, 350, 250, 150, 100, 50, 5)
    rec = dict(K=[], K_off=[0], delta=[], delta_off=[0], A=[], E=[], min_margin=[], min_margin_k=[], min_margin_logits=[])
    prev = np.zeros(0, np.int64)
    extra = {}
    import time
    t0 = time.time()

    def spy(batched_graph, t_node, t_edge):
        nonlocal prev
        seen.pop('logits', None)
        out = BinomialDiffusionActive.p_sample(model, batched_graph, t_node, t_edge)
        t = int(t_node[0])
        K = batched_graph.active_node_indices.numpy().astype(np.int32)
        A = int(batched_graph.active_edge_indices.shape[0])
        state = ones_idx(out[1])
        rec['K'].append(K); rec['K_off'].append(rec['K_off'][-1] + len(K))
        d = np.setxor1d(prev, state).astype(np.int64)
        rec['delta'].append(d); rec['delta_off'].append(rec['delta_off'][-1] + len(d))
        rec['A'].append(A); rec['E'].append(len(state))
        mm, mk, ml = np.float32(np.inf), -1, np.zeros(2, np.float32)
        if A:
            lg = seen['logits']
            lp = torch.log_softmax(lg, dim=1)
            u = last['u']
            assert tuple(u.shape) == (A, 2)
            gum = -torch.log(-torch.log(u + 1e-30) + 1e-30)
            s = gum + lp
            margin = (s[:, 1] - s[:, 0])
            # the reference's own draw, recomputed: must reproduce the bits it wrote
            bits = out[1][batched_graph.active_edge_indices].argmax(1)
            assert torch.equal((margin > 0).long(), bits)
            k = int(margin.abs().argmin())
            mm, mk, ml = np.float32(margin[k]), k, lg[k].numpy()
            if t in LOGIT_STEPS:
                extra[f'logits_{t}'] = lg.numpy()
        rec['min_margin'].append(mm); rec['min_margin_k'].append(mk); rec['min_margin_logits'].append(ml)
        prev = state
        if t % 32 == 0:
            print(f'    t={t} K={len(K)} A={A} E={len(state)}  {time.time() - t0:.0f}s', flush=True)
        return out
    model.p_sample = spy
    import io
    import contextlib
    with torch.no_grad(), contextlib.redirect_stdout(sys.stderr):
        res = model.sample(len(graphs))
    steps = len(rec['A'])
    assert steps == T
    save(f'trajbig_{case}.npz', T=T, max_degree=dmax, uniform_seed=useed + 3,
         K=np.concatenate(rec['K']), K_off=np.array(rec['K_off'], np.int64),
         delta=np.concatenate(rec['delta']), delta_off=np.array(rec['delta_off'], np.int64),
         A=np.array(rec['A'], np.int64), E=np.array(rec['E'], np.int64),
         min_margin=np.array(rec['min_margin'], np.float32), min_margin_k=np.array(rec['min_margin_k'], np.int64),
         min_margin_logits=np.stack(rec['min_margin_logits']).astype(np.float32),
         edge_index=res.edge_index, **extra)
    print(f'    final edges {res.edge_index.shape[1]}, decisions {int(np.sum(rec["A"]))}, '
          f'smallest |margin| {float(np.min(np.abs(rec["min_margin"]))):.3e}')


def gen_stats():
    """compute_graph_statistics of the reference (eval_utils/graph_statistics.py:249-285) on the fixture graphs, and
    NetworkEvaluator.evaluate's NMAE dict (datasets/evaluator.py:32-37; that module needs dgl to import, so its two
    lines are applied here to the reference statistics).  Stand-ins: `powerlaw` is absent - its Fit for this call is
    restated (PARITY UNPINNED for power_law_exp); `nx.from_scipy_sparse_matrix` (removed in networkx 3) is aliased to
    its successor so that LCC / triangles / assortativity take their intended path instead of the bare `except`."""
    print('graph statistics')
    import networkx as nx
    pl = types.ModuleType('powerlaw')

    class Fit:
        def __init__(self, data, xmin=None, verbose=False):
            d = np.asarray(data, dtype=np.float64)
            d = d[d > 0]
            d = d[d >= xmin]
            self.power_law = types.SimpleNamespace(alpha=1.0 + len(d) / np.sum(np.log(d / xmin)))
    pl.Fit = Fit
    sys.modules['powerlaw'] = pl
    if not hasattr(nx, 'from_scipy_sparse_matrix'):
        nx.from_scipy_sparse_matrix = nx.from_scipy_sparse_array
    import warnings
    from eval_utils.graph_statistics import compute_graph_statistics
    keys = None
    rows = []
    for g in cases.stats_graphs():
        G = nx.Graph()
        G.add_nodes_from(range(g['n']))
        G.add_edges_from(np.asarray(g['edges']).reshape(-1, 2).tolist())
        A = nx.to_scipy_sparse_array(G).astype(np.float64) if g['n'] > 0 else None
        with warnings.catch_warnings(), np.errstate(all='ignore'):
            warnings.simplefilter('ignore')
            st = compute_graph_statistics(A)
        keys = keys or list(st.keys())
        rows.append([float(np.asarray(st[k]).reshape(-1)[0]) if np.size(st[k]) else np.nan for k in keys])
        print('   ', g['n'], {k: round(v, 4) for k, v in zip(keys, rows[-1])})
    rows = np.array(rows, dtype=np.float64)
    # NetworkEvaluator: reference = the polblogs-shaped graph (last), targets = the three graphs before it
    ref, tg = rows[-1], rows[-4:-1]
    with np.errstate(all='ignore'):
        mean = tg.mean(0)
        nmae = np.abs((mean - ref) / ref)
    save('stats.npz', keys=np.array(keys), values=rows, nmae=nmae, nmae_mean=mean)


def gen_sampler():
    """EmpiricalEmptyGraphGenerator of the reference (datasets/data_utils.py:78-147) under a seed: the ORDER and
    COUNT of its RNG draws is the contract (one torch multinomial for all sizes, then one `random.choice` per graph).
    The module needs torch_geometric: Data / Batch are attribute bags here (clone, from_data_list = concatenation of
    the per-graph tensors), which is all `sample()` asks of them."""
    print('empty-graph sampler')
    import copy
    import random
    pyg = sys.modules['torch_geometric']
    pyg.data = types.ModuleType('torch_geometric.data')
    pyg.transforms = types.ModuleType('torch_geometric.transforms')
    pyg.utils.to_dense_adj = None
    pyg.utils.from_networkx = None

    class Data:
        def clone(self):
            return copy.deepcopy(self)

    class Batch(Data):
        @staticmethod
        def from_data_list(ds):
            b = Batch()
            b.num_graphs = len(ds)
            b.degree = torch.cat([d.degree for d in ds])
            b.node_attr = torch.cat([d.node_attr for d in ds])
            b.full_edge_attr = torch.cat([d.full_edge_attr for d in ds])
            b.batch = torch.cat([torch.full((d.num_nodes,), i, dtype=torch.long) for i, d in enumerate(ds)])
            return b
    pyg.data.Data, pyg.data.Batch = Data, Batch
    for mod in (pyg.data, pyg.transforms):
        sys.modules[mod.__name__] = mod
    from datasets.data_utils import EmpiricalEmptyGraphGenerator
    train = []
    for g in cases.sampler_train_graphs():
        d = Data()
        d.num_nodes = g['n']
        e = torch.as_tensor(np.asarray(g['edges']).reshape(-1, 2), dtype=torch.long).t()
        d.edge_index = torch.cat([e, e.flip(0)], dim=1)          # bi-directional, as pyg.utils.from_networkx gives it
        train.append(d)
    gen = EmpiricalEmptyGraphGenerator(train, degree=True)
    torch.manual_seed(cases.SAMPLER_SEED)
    random.seed(cases.SAMPLER_SEED)
    b1 = gen.sample(cases.SAMPLER_DRAWS)
    b2 = gen.sample(7)                                            # a second call continues both RNG streams
    save('sampler.npz', size_dist=gen.empirical_graph_size_dist, min_node=gen.min_node, max_node=gen.max_node,
         nodes_per_graph_1=b1.nodes_per_graph, edges_per_graph_1=b1.edges_per_graph, degree_1=b1.degree,
         nodes_per_graph_2=b2.nodes_per_graph, edges_per_graph_2=b2.edges_per_graph, degree_2=b2.degree,
         degree_dtype=str(b1.degree.dtype))
    print('    sizes', b1.nodes_per_graph.tolist())


def synth_deg(g):
    from edgehip import synth
    return synth.degrees(g['n'], g['edges']).astype(np.float32)


if __name__ == '__main__':
    torch.manual_seed(0)
    real_rand_like = torch.rand_like
    if len(sys.argv) > 1 and sys.argv[1] == 'dropout':
        try:
            gen_loss('ragged', [0, 7, 47, 12, 1, 20], tag='_dp', dp=0.1)
            gen_loss('polblogs1', [100], tag='_dp', dp=0.1)
        finally:
            torch.rand_like = real_rand_like
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'sampler':
        gen_sampler()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'stats':
        gen_stats()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'trajbig':
        try:
            gen_traj_big('polblogs1')
        finally:
            torch.rand_like = real_rand_like
        sys.exit(0)
    try:
        gen_schedule()
        gen_posterior()
        gen_step('toy', 5)
        gen_step('ragged', 9)
        gen_step('polblogs1', 100)
        gen_loss('toy', [3, 0])
        gen_loss('toy', [6, 4], tag='_mid')
        gen_loss('ragged', [0, 7, 47, 12, 1, 20])
        gen_loss('polblogs1', [100])
        gen_loss('ragged', [0, 7, 47, 12, 1, 20], tag='_dp', dp=0.1)      # README.md:49,79: --dp_rate 0.1
        gen_loss('polblogs1', [100], tag='_dp', dp=0.1)
        gen_traj('toy')
        gen_traj('ragged')
        gen_traj_big('polblogs1')
        gen_stats()
        gen_sampler()
    finally:
        torch.rand_like = real_rand_like
    print('golden vectors written; oracle == reference on every case')
