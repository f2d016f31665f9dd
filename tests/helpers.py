"""Shared test helpers: golden loading, numpy Philox, model/batch construction."""
import os
from types import SimpleNamespace

import numpy as np
import torch

import cases
from oracle import edge_oracle as orc

GOLDEN = cases.GOLDEN_DIR


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


# ---- Philox4x32-10 in numpy (checked against the Random123 known-answer vectors) ----------
def philox4x32(seed, c0, c1, c2, c3):
    c = [np.asarray(x, dtype=np.uint64) & np.uint64(0xFFFFFFFF) for x in np.broadcast_arrays(c0, c1, c2, c3)]
    k0, k1 = np.uint64(seed & 0xFFFFFFFF), np.uint64((seed >> 32) & 0xFFFFFFFF)
    M0, M1, MASK = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        c = [(p1 >> np.uint64(32)) ^ c[1] ^ k0, p1 & MASK, (p0 >> np.uint64(32)) ^ c[3] ^ k1, p0 & MASK]
        k0 = (k0 + np.uint64(0x9E3779B9)) & MASK
        k1 = (k1 + np.uint64(0xBB67AE85)) & MASK
    return [x.astype(np.uint32) for x in c]


def u24(x):
    return ((x >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)).astype(np.float32)


STREAM_NODE, STREAM_PAIR, STREAM_QTM1 = 0, 1, 2


def rng_node_uniforms(seed, nodes_per_graph, t, gid_base=0):
    """[Nn,2] uniforms the device draws for the active-node selection at step t."""
    out = []
    for b, n in enumerate(nodes_per_graph):
        x = philox4x32(seed, np.arange(n), gid_base + b, 0, t * 8 + STREAM_NODE)
        out.append(np.stack([u24(x[0]), u24(x[1])], 1))
    return torch.from_numpy(np.concatenate(out))


def rng_pair_uniforms(seed, pair_idx_local, graph_ids, t, stream=STREAM_PAIR, gid_base=0, four=False):
    pidx = np.asarray(pair_idx_local, dtype=np.uint64)
    x = philox4x32(seed, pidx & np.uint64(0xFFFFFFFF), pidx >> np.uint64(32), np.asarray(graph_ids) + gid_base,
                   np.asarray(t) * 8 + stream)
    cols = [u24(x[0]), u24(x[1])] + ([u24(x[2]), u24(x[3])] if four else [])
    return torch.from_numpy(np.stack(cols, 1))


# ---- product model / batch builders (GPU tests) -----------------------------------------------
def make_args(T, max_degree, device='cuda', heads=cases.HEADS):
    return SimpleNamespace(final_prob_node=None, final_prob_edge=[1 - 1e-12, 1e-12], num_node_classes=2,
                           num_edge_classes=2, arch='TGNN_degree_guided', parametrization='xt_prescribed_st',
                           loss_type='vb_ce_xt_prescribred_st', max_degree=max_degree, diffusion_dim=cases.DIM,
                           diffusion_steps=T, num_heads=list(heads), dp_rate=0., norm='None', degree=True,
                           augmented_feature_dict={}, has_node_feature=False, noise_schedule='linear',
                           sample_time_method='importance', device=device)


def build_model(params, T, max_degree, sampler=None, device='cuda', schedule='linear'):
    import edgehip
    a = make_args(T, max_degree, device)
    a.noise_schedule = schedule
    m = edgehip.get_model(a, sampler)
    m._denoise_fn.load_state_dict(params, strict=True)
    return m.to(device).eval()


def graph_batch(graphs, with_edges=True, degree='true', device='cuda', degree_dtype=None, gid_base=0):
    import edgehip
    from edgehip import synth
    degs = [synth.degrees(g['n'], g['edges']) for g in graphs] if degree == 'true' else None
    return edgehip.GraphBatch([g['n'] for g in graphs], degree=degs,
                              edges=[g['edges'] for g in graphs] if with_edges else None, device=device,
                              degree_dtype=degree_dtype, gid_base=gid_base)


def set_state_from_pair_indices(batch, idx):
    """Load a dense pair-index list (reference enumeration) into batch.adj."""
    import torch
    fei = orc.make_batch([dict(n=int(n), edges=None) for n in batch.host.nodes_per_graph]).full_edge_index
    e = fei[:, torch.as_tensor(idx, dtype=torch.long)].t().contiguous().numpy()
    batch.adj.zero_()
    batch.set_edges(batch.adj, e)


def rel_err(a, b):
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    if b.numel() == 0:
        return 0.0
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-30))


def check_logits(got, ref, tol=1e-4, what='logits'):
    """Logit parity, three ways (the north star's "1e-4 relative"):
      max-norm      max|got-ref| / max|ref| < tol                      (what `rel_err` reports)
      element-wise  |got-ref| <= tol*|ref| + 2e-6 for EVERY element     (2e-6: a few fp32 ulps of the O(1) pre-activation
                    sums; a logit near zero cannot be held to a relative bound of itself)
      reported      the worst |got-ref|/|ref| over the elements with |ref| > 1e-2 max|ref|
    Returns the three figures; asserts the first two."""
    a, b = torch.as_tensor(got).detach().double().cpu(), torch.as_tensor(ref).detach().double().cpu()
    if b.numel() == 0:
        return dict(max_norm=0.0, elementwise_ok=True, worst_elementwise=0.0)
    d = (a - b).abs()
    scale = max(float(b.abs().max()), 1e-30)
    max_norm = float(d.max()) / scale
    big = b.abs() > 1e-2 * scale
    worst = float((d[big] / b.abs()[big]).max()) if bool(big.any()) else 0.0
    viol = d > tol * b.abs() + 2e-6
    assert max_norm < tol, f'{what}: max-norm relative error {max_norm:.3e}'
    assert not bool(viol.any()), f'{what}: {int(viol.sum())} elements beyond {tol}*|ref| + 2e-6 (worst abs {float(d.max()):.3e})'
    assert worst < tol, f'{what}: element-wise relative error {worst:.3e} on an element above 1% of the scale'
    return dict(max_norm=max_norm, elementwise_ok=True, worst_elementwise=worst)


def kl_prior_fp64(o, bt):
    """The reference's prior KL (diffusion_binomial_vanilla.py:116-136) with its per-element fp32 values - computed
    exactly as the reference computes each element - but SUMMED in fp64.  The reference adds them in fp32 (scatter),
    which drifts by up to ~1e-4 relative at P = 746k identical terms; the product forms count x value in fp64."""
    s = o.sched
    tn = (o.T - 1) * torch.ones(bt.num_nodes).long()
    te = (o.T - 1) * torch.ones(int(bt.edges_per_graph.sum())).long()
    qn = orc.q_pred(s, bt.log_node_attr, tn, s.log_final_prob_node)
    qe = orc.q_pred(s, bt.log_full_edge_attr, te, s.log_final_prob_edge)
    kn = (qn.exp() * (qn - s.log_final_prob_node * torch.ones_like(qn))).sum(dim=1).double()
    ke = (qe.exp() * (qe - s.log_final_prob_edge * torch.ones_like(qe))).sum(dim=1).double()
    out = torch.zeros(bt.num_graphs, dtype=torch.float64)
    out.index_add_(0, bt.batch, kn)
    out.index_add_(0, bt.batch[bt.full_edge_index[0]], ke)
    return out
