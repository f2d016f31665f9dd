"""GPU: a whole polblogs-scale reference trajectory (T = 512, 53 k Gumbel decisions on network logits over 512 steps) replayed
step by step against `tests/golden/trajbig_polblogs1.npz` (written by make_golden.py from the imported reference).

What is asserted and what is measured:
  * given the reference state of a step, the degrees / active-node list are bit-exact at EVERY step (they come from
    integer popcounts, the host-built posterior table and the host-built Gumbel LUT);
  * the sampled pair bits are bit-exact given logits + uniforms, but the logits are fp32 network outputs that differ
    from the CPU reference's in the last bits (~5e-7): a decision whose margin |g1+lp1 - g0-lp0| is below that noise
    can come out the other way.  The test runs FREE until the first such step, records it, re-synchronises the state
    to the reference and goes on, so every later step is again "same state in, same decision out?".  It asserts that
    every differing decision has a margin below 1e-4 (i.e. explained by logit rounding, not by a defect), that there
    are at most a handful, and writes what it saw to gpurun_out/trajectory_replay.json.
Uniforms are redrawn from the fixture's seed in the reference's consumption order."""
import json
import os

import numpy as np
import pytest
import torch

import cases
import helpers as H
from oracle import edge_oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _pair_idx(ei, n):
    i, j = ei[0].astype(np.int64), ei[1].astype(np.int64)
    return i * (2 * n - i - 1) // 2 + (j - i - 1)


def _pairs_of(idx, n):
    """inverse of the triu enumeration for an int64 array"""
    idx = np.asarray(idx, dtype=np.int64)
    b = 2.0 * n - 1.0
    i = np.floor((b - np.sqrt(b * b - 8.0 * idx)) / 2.0).astype(np.int64)
    off = lambda r: r * (2 * n - r - 1) // 2
    i = np.where(off(i) > idx, i - 1, i)
    i = np.where(off(i + 1) <= idx, i + 1, i)
    j = idx - off(i) + i + 1
    return np.stack([i, j], 1)


def test_polblogs_trajectory_replay_against_the_reference():
    g = H.golden('trajbig_polblogs1.npz')
    builder, T, wseed, useed = cases.CASES['polblogs1']
    graphs = builder()
    n = graphs[0]['n']
    Nn, P = n, n * (n - 1) // 2
    assert int(g['T']) == T == 512
    model = H.build_model(orc.random_tgnn_params(wseed), T, int(g['max_degree']))
    b = H.graph_batch(graphs, with_edges=False)
    rng = np.random.default_rng(int(g['uniform_seed']))
    cases.uniforms(rng, Nn, 2); cases.uniforms(rng, P, 2)            # the draws of _prepare_data_for_sampling (x_T = empty)
    K, K_off, delta, d_off = g['K'], g['K_off'], g['delta'], g['delta_off']
    ref_state = np.zeros(0, np.int64)                                 # reference state BEFORE the step
    b.adj.zero_()
    synced = True
    first_free_divergence = None
    diffs, decisions, logit_err = [], 0, {}
    for s, t in enumerate(range(T - 1, -1, -1)):
        u_act = cases.uniforms(rng, Nn, 2)
        drawn = {}

        def pair_u(A):
            cases.uniforms(rng, Nn, 2)                               # the node draw (always class 0) still consumes
            drawn['u'] = cases.uniforms(rng, A, 2)
            return drawn['u'].cuda()
        want = {}
        model.p_sample(b, t, u_act.cuda(), pair_u, None, None, want)
        ani, _ = model.materialize_actives(b)
        refK = K[K_off[s]:K_off[s + 1]]
        assert np.array_equal(ani.cpu().numpy(), refK), f't={t}: active nodes differ given the same state'
        A = int(g['A'][s])
        assert (len(drawn['u']) == A) if A else ('u' not in drawn)
        decisions += A
        ref_after = np.setxor1d(ref_state, delta[d_off[s]:d_off[s + 1]])
        ei, _ = b.extract_edges()
        got = np.sort(_pair_idx(ei.cpu().numpy(), n)) if ei.shape[1] else np.zeros(0, np.int64)
        if f'logits_{t}' in g.files:
            logit_err[t] = (H.rel_err(want['logits'], g[f'logits_{t}']),
                            float((want['logits'].cpu() - torch.from_numpy(g[f'logits_{t}'])).abs().max()))
        if not np.array_equal(got, ref_after):
            wrong = np.setxor1d(got, ref_after)
            if first_free_divergence is None:
                first_free_divergence = t
            # margins of the differing decisions from THIS side's logits and the shared uniforms
            lg = want['logits'].cpu().double()
            lp = torch.log_softmax(lg, 1)
            u = drawn['u'].double()
            gum = -torch.log(-torch.log(u + 1e-30) + 1e-30)
            margin = ((gum[:, 1] + lp[:, 1]) - (gum[:, 0] + lp[:, 0])).numpy()
            # active pair k <-> pair index: enumerate the pairs of the active list
            kk = refK.astype(np.int64)
            ai, aj = np.triu_indices(len(kk), 1)
            aidx = kk[ai] * (2 * n - kk[ai] - 1) // 2 + (kk[aj] - kk[ai] - 1)
            for w in wrong:
                k = int(np.nonzero(aidx == w)[0][0])
                entry = dict(t=t, pair=int(w), k=k, margin_here=float(margin[k]))
                if int(g['min_margin_k'][s]) == k:
                    entry['margin_reference'] = float(g['min_margin'][s])
                    entry['logit_abs_err'] = float(np.abs(want['logits'][k].cpu().numpy() - g['min_margin_logits'][s]).max())
                diffs.append(entry)
            # re-synchronise to the reference state
            b.adj.zero_()
            if len(ref_after):
                b.set_edges(b.adj, _pairs_of(ref_after, n))
        assert int(g['E'][s]) == len(ref_after)
        ref_state = ref_after
    report = dict(steps=T, decisions=decisions, first_step_with_a_different_decision=first_free_divergence,
                  different_decisions=diffs, smallest_reference_margin=float(np.min(np.abs(g['min_margin']))),
                  logits_vs_reference={str(t): dict(max_norm_rel=e[0], max_abs=e[1]) for t, e in logit_err.items()},
                  final_edges=int(len(ref_state)))
    out = os.path.join(ROOT, 'gpurun_out')
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, 'trajectory_replay.json'), 'w') as fh:
        json.dump(report, fh, indent=1)
    print(json.dumps(report))
    assert decisions == int(g['A'].sum()) and decisions > 50_000
    assert all(e[0] < 1e-4 for e in logit_err.values()), logit_err
    assert len(diffs) <= 8, diffs
    assert all(abs(d['margin_here']) < 1e-4 for d in diffs), diffs
    # the final state of the (re-synchronised) replay is the reference's final edge list
    fin = np.sort(_pair_idx(g['edge_index'], n))
    assert np.array_equal(fin, ref_state)


def test_polblogs_trajectory_through_the_fused_step_equals_the_reference():
    """The same 512 steps through edgehip_sample_step (the entry point the sampler and bench.py run: one call per
    step, no host sync), uniforms supplied: per-step edge counts and the final edge list equal the reference's.

    This run is FREE (no re-synchronisation), so it holds only while no decision's margin is inside the fp32 rounding
    noise of the logits (~2e-7; the smallest reference margin on this trajectory is 4.5e-6).  A kernel change that
    reorders a sum can legitimately flip a near-tie: a mismatch is a failure only if the closest reference decision of
    the first differing step has a margin above 1e-4 - otherwise the test skips and points at the replay test above,
    which re-synchronises and checks each differing decision's margin."""
    import ctypes as C
    from edgehip import capi
    from edgehip.model import _state, _workspace
    g = H.golden('trajbig_polblogs1.npz')
    builder, T, wseed, useed = cases.CASES['polblogs1']
    graphs = builder()
    n = graphs[0]['n']
    Nn, P = n, n * (n - 1) // 2
    model = H.build_model(orc.random_tgnn_params(wseed), T, int(g['max_degree']))
    b = H.graph_batch(graphs, with_edges=False)
    net = model._denoise_fn
    Pp, folded = net.c_params()
    tabs = model._tables(b.device, int(b.degree_i32.max().item()))
    ws = _workspace(b, Pp)
    counts = _state(b)['counts']
    rng = np.random.default_rng(int(g['uniform_seed']))
    cases.uniforms(rng, Nn, 2); cases.uniforms(rng, P, 2)
    b.adj.zero_()
    dummy = torch.zeros((1, 2), dtype=torch.float32, device='cuda')
    keep = []

    def mismatch(s, what):
        """s: the first step whose outcome differs (the state is compared after EVERY step)."""
        tight = float(abs(g['min_margin'][s])) if int(g['A'][s]) else float('inf')
        if tight < 1e-4:
            pytest.skip(f'{what}: the closest reference decision of this step has margin {tight:.2e} - a near-tie may '
                        'have flipped; test_polblogs_trajectory_replay_against_the_reference decides')
        pytest.fail(f'{what} (closest reference margin of the step {tight:.2e}: not a near-tie)')
    for s, t in enumerate(range(T - 1, -1, -1)):
        u_act = cases.uniforms(rng, Nn, 2).cuda()
        A = int(g['A'][s])
        u_pair = dummy
        if A:
            cases.uniforms(rng, Nn, 2)
            u_pair = cases.uniforms(rng, A, 2).cuda()
        keep = [u_act, u_pair]
        capi.check(capi.lib.edgehip_sample_step(C.byref(b.layout), C.byref(Pp), folded.data_ptr(), b.adj.data_ptr(),
                                                b.degree_i32.data_ptr(), t, tabs['post'].data_ptr(), tabs['nmax'],
                                                u_act.data_ptr(), u_pair.data_ptr(), None, capi.ptr(tabs['lut']),
                                                counts.data_ptr(), ws.data_ptr(), ws.numel(), capi.stream_ptr()), 'sample_step')
        # compared after every step: the first differing step names the decisions to look at
        c = counts.cpu().numpy()
        if int(c[2:4].view(np.int64)[0]) != A:
            mismatch(s, f't={t}: the fused step saw a different number of active pairs')
        _, sl = b.extract_edges()
        if int(sl[-1]) != int(g['E'][s]):
            mismatch(s, f't={t}: edge count differs')
    ei, _ = b.extract_edges()
    if not np.array_equal(ei.cpu().numpy(), g['edge_index']):
        pytest.fail('final edge list differs although every per-step edge count matched')
