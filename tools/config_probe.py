"""Throughput of the BASELINE configs that bench.py does not time (they are parity cases there): sampling on
cora-, Homo_sapiens- and SBM-20k-shaped graphs (T = 512), Ego-shaped training (T = 128, batch 8) and sampling.
Synthetic graphs of the published shapes, random-init network.  Prints one JSON object.
    python tools/config_probe.py [out.json]"""
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


def sample_rate(n, edges, B, T, reps=1):
    import edgehip
    from edgehip import synth
    model, deg = bench.make_model(n, edges, 'cuda', T=T)
    batch = edgehip.GraphBatch([n] * B, degree=[deg] * B, device='cuda')
    out = model.sample_batch(batch, seed=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(reps):
        out = model.sample_batch(batch, seed=2 + i)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / reps
    e = int(out._slice_dict['edge_index'][-1].item()) / B
    print(f'  n {n} B {B} T {T}: {el:.2f} s per pass', flush=True)
    return {'nodes': n, 'true_edges': int(len(edges)), 'graphs_per_pass': B, 'diffusion_steps': T, 's_per_pass': el,
            'graphs_per_s': B / el, 'us_per_reverse_step': 1e6 * el / T, 'mean_edges_per_sampled_graph': e}


def main():
    import edgehip
    from edgehip import synth
    res = {'note': 'one MI355X, synthetic graphs of the published shapes, random-init TGNN_degree_guided (dim 64, heads '
                   '8/8/8/8/1); not bench.py lines (BASELINE quotes its metric on polblogs)'}
    # Ego: batch 8 graphs of 50..400 nodes, T = 128: training step and sampling of 64
    gs = synth.ego_like(16, seed=0)
    T = 128
    dmax = max(int(synth.degrees(n, e).max()) for n, e in gs)
    from types import SimpleNamespace
    sampler = edgehip.EmpiricalEmptyGraphGenerator(gs, degree=True, device='cuda')
    args = SimpleNamespace(final_prob_node=None, final_prob_edge=[1 - 1e-12, 1e-12], num_node_classes=2,
                           num_edge_classes=2, arch='TGNN_degree_guided', parametrization='xt_prescribed_st',
                           loss_type='vb_ce_xt_prescribred_st', max_degree=dmax, diffusion_dim=64,
                           diffusion_steps=T, num_heads=bench.HEADS, dp_rate=0., norm='None', degree=True,
                           augmented_feature_dict={}, has_node_feature=False, noise_schedule='linear',
                           sample_time_method='importance', device='cuda')
    torch.manual_seed(0)
    model = edgehip.get_model(args, sampler).to('cuda')
    model.rng_seed = 3
    for name, reps in (('ego_T128_train_batch8', 1), ('ego_T128_train_batch512', 64)):
        sel = [gs[i % len(gs)] for i in range(8)] * reps
        tb = edgehip.GraphBatch([n for n, _ in sel], degree=[synth.degrees(n, e) for n, e in sel], edges=[e for _, e in sel],
                                device='cuda', degree_dtype=np.int64)
        opt = edgehip.make_optimizer(model)
        state = {}
        for _ in range(5):
            edgehip.train_step(model, tb, opt, clip_value=1.0, state=state)
        torch.cuda.synchronize()
        # per-step times (t, hence the number of active pairs and the sizes torch allocates, differs from step to
        # step: the first steps at a new size pay the caching allocator); median and mean are both kept
        ts = []
        for _ in range(20 if reps == 1 else 12):
            t0 = time.perf_counter()
            edgehip.train_step(model, tb, opt, clip_value=1.0, state=state)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        el = float(np.median(ts))
        res[name] = {'graphs_per_step': len(sel), 'nodes_per_step': int(sum(n for n, _ in sel)), 'ms_per_step': 1e3 * el,
                     'ms_per_step_mean': 1e3 * float(np.mean(ts)), 'ms_per_step_max': 1e3 * float(np.max(ts)),
                     'graphs_per_s': len(sel) / el}
        print(f'  {name}: {1e3 * el:.2f} ms', flush=True)
    model.eval()
    torch.manual_seed(1)
    out = model.sample(64)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = model.sample(64)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    res['ego_T128_sample64'] = {'s_per_pass': el, 'graphs_per_s': 64 / el,
                                'mean_edges_per_sampled_graph': int(out._slice_dict['edge_index'][-1].item()) / 64}
    torch.cuda.empty_cache()
    if os.environ.get('PROBE_ONLY', '') != 'ego':
        n, e = synth.cora_like(0)
        res['cora_T512'] = sample_rate(n, e, 64, 512)
        n, e = synth.homo_sapiens_like(0)
        res['homo_sapiens_T512'] = sample_rate(n, e, 32, 512)
        n, e = synth.sbm_graph(20, 1000, 0.02, 5e-4, 0)
        res['sbm20k_T512'] = sample_rate(n, e, 4, 512)
    s = json.dumps(res)
    print(s)
    if len(sys.argv) > 1:
        open(sys.argv[1], 'w').write(s + '\n')


if __name__ == '__main__':
    main()
