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


def build_model(params, T, max_degree, sampler=None, device='cuda'):
    import edgehip
    m = edgehip.get_model(make_args(T, max_degree, device), sampler)
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
