"""Sampling throughput with a TRAINED network instead of the random-init one bench.py uses: the work of a reverse step
depends on how many nodes the network activates (`trajectory`).  Trains on the polblogs-shaped graph for
a few thousand steps (seconds), then times sample_batch of 64 graphs and prints the trajectory profile.
    python tools/trained_probe.py [train_steps=3000] [out.json]"""
import json
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'graph-generation-edge_amd'))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402


def main():
    import edgehip
    from edgehip import synth
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    n, edges = synth.polblogs_like(0)
    model, deg = bench.make_model(n, edges, 'cuda')
    tb = edgehip.GraphBatch([n] * 4, degree=[deg] * 4, edges=[edges] * 4, device='cuda', degree_dtype=np.int64)
    model.rng_seed = 1
    torch.manual_seed(0)
    opt = edgehip.make_optimizer(model)
    state = {}
    t0 = time.perf_counter()
    bpd = []
    for i in range(steps):
        b = edgehip.train_step(model, tb, opt, clip_value=1.0, state=state)
        if i % 250 == 0 or i == steps - 1:
            bpd.append((i, float(b)))
    torch.cuda.synchronize()
    train_s = time.perf_counter() - t0
    model.eval()
    B = 64
    batch = edgehip.GraphBatch([n] * B, degree=[deg] * B, device='cuda')
    out = model.sample_batch(batch, seed=1)
    torch.cuda.synchronize()
    reps = 3
    by_parts = {}
    for parts in (1, 2, 3):
        t0 = time.perf_counter()
        for i in range(reps):
            out = model.sample_batch(batch, seed=2 + i, parts=parts)
        torch.cuda.synchronize()
        by_parts[parts] = B * reps / (time.perf_counter() - t0)
    el = B / by_parts[3]
    from edgehip import capi
    stats = torch.zeros((bench.T_STEPS, 8), dtype=torch.int64, device='cuda')
    torch.cuda.synchronize()
    capi.lib.edgehip_prof_enable(1)
    out = model.sample_batch(batch, seed=2 + reps - 1, parts=1, step_stats=stats)
    torch.cuda.synchronize()
    prof = capi.prof_collect()
    capi.lib.edgehip_prof_enable(0)
    tot = sum(v[0] for v in prof.values())
    kern = {k: {'ms': round(v[0], 2), 'avg_us': round(1e3 * v[0] / v[1], 2), 'share': round(v[0] / tot, 3)}
            for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])}
    st = stats.cpu().numpy()[::-1]
    idx = list(range(0, bench.T_STEPS, 64)) + [bench.T_STEPS - 1]
    res = {'train_steps': steps, 'train_s': train_s, 'bpd': bpd, 'graphs_per_s': B / el, 's_per_pass': el,
           'true_edges': int(len(edges)), 'mean_edges_per_sampled_graph': int(out._slice_dict['edge_index'][-1].item()) / B,
           'total_pair_decisions': int(st[:, 1].sum()),
           't': [bench.T_STEPS - 1 - i for i in idx], 'active_nodes_K': [int(st[i, 0]) for i in idx],
           'active_pairs_A': [int(st[i, 1]) for i in idx], 'edges_E_t': [int(st[i, 6] // 2) for i in idx],
           'idle_graph_step_fraction': float(1.0 - st[:, 2].sum() / (B * bench.T_STEPS)), 'kernels_one_stream': kern, 'graphs_per_s_by_streams': by_parts}
    s = json.dumps(res)
    print(s)
    if len(sys.argv) > 2:
        open(sys.argv[2], 'w').write(s + '\n')


if __name__ == '__main__':
    main()
