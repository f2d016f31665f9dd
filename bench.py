"""bench.py - sampled graphs/sec of the degree-guided reverse process, polblogs shape, T = 512.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one batch: model.sample() of `--graphs` (default 64)
polblogs-shaped graphs through all 512 reverse steps, state resident in HBM, ending with the edge
lists on the device.  Every rank samples its own `--graphs` graphs (weak scaling, no data-path
collective; graph global ids key the RNG so results do not depend on the sharding).
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'graph-generation-edge_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

T_STEPS = 512
HEADS = [8, 8, 8, 8, 1]
MFMA_F32_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md, Peak FP32 (matrix)
MFMA_BF16_PEAK_TFLOPS = 2500.0    # MI355X_MICROARCH.md, BF16 dense
BF16_PRODUCTS_PER_FP32 = 6        # csrc/gemm.h: exact 3-way bf16 split of both operands, six products kept
HBM_PEAK_GBPS = 8000.0            # MI355X_MICROARCH.md, HBM3E peak
SURVEY_GB_PER_GRAPH = 40.7        # algorithmic HBM bytes per sampled polblogs graph


def make_model(n, edges, device, T=T_STEPS):
    from types import SimpleNamespace
    import edgehip
    from edgehip import synth
    d = synth.degrees(n, edges)
    sampler = edgehip.EmpiricalEmptyGraphGenerator([(n, edges)], degree=True, device=device)
    args = SimpleNamespace(final_prob_node=None, final_prob_edge=[1 - 1e-12, 1e-12], num_node_classes=2,
                           num_edge_classes=2, arch='TGNN_degree_guided', parametrization='xt_prescribed_st',
                           loss_type='vb_ce_xt_prescribred_st', max_degree=int(d.max()), diffusion_dim=64,
                           diffusion_steps=T, num_heads=HEADS, dp_rate=0., norm='None', degree=True,
                           augmented_feature_dict={}, has_node_feature=False, noise_schedule='linear',
                           sample_time_method='importance', device=device)
    torch.manual_seed(0)
    model = edgehip.get_model(args, sampler).to(device).eval()
    return model, d


def kernel_work(name, Nn, launches_per_pass):
    """Algorithmic flops / bytes of ONE launch of a kernel class at this workload (DESIGN.md 'Kernels').
    Per-layer kernels run once per layer with different head counts; their per-launch figure is the
    average over the 5 layers.  Flops are those of the FOLDED formulation the library executes
    (keys / values never formed), i.e. the lower, honest count - not the reference's 2.6 MFLOP/node."""
    hs = HEADS
    if name == 'layer':          # fused layer: q~ = h Mcat, conv = [z|h] Ocat, GRU (6 x 64x64), ctx 64->256
        fl = np.mean([2.0 * Nn * (64 * 64 * h + (64 * h + 64) * 64 + 6 * 64 * 64 + 64 * 256) for h in hs])
        # HBM: read h row + bit row (~160 B) , write h' row + per-tile context sums; q~/z stay in LDS,
        # neighbour rows and weights are L2-resident re-reads
        by = Nn * (256 + 160 + 256) + (Nn / 32) * 1024
    else:                        # heavy-row attention etc. are data-dependent: no fixed per-launch figure
        return None
    return fl, by


def pmc_traffic(kernel_class):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (rocprofv3 cannot run
    inside this process): profiles/r01_pmc_traffic.json, made by tools/pmc_traffic.py from two
    `rocprofv3 --pmc` runs of this same script; None when the file or the kernel is absent."""
    path = os.path.join(ROOT, 'profiles', 'r01_pmc_traffic.json')
    pref = {'layer': 'k_layer'}.get(kernel_class)
    if pref is None or not os.path.exists(path):
        return None, None
    with open(path) as fh:
        kk = json.load(fh)['kernels']
    sel = [v for k, v in kk.items() if k.startswith(pref)]
    if not sel:
        return None, None
    n = sum(v['launches'] for v in sel)
    return sum(v['hbm_bytes_per_launch'] * v['launches'] for v in sel) / n, 'profiles/r01_pmc_traffic.json'


def cpu_baseline(model, n, edges, deg, budget_s=20.0):
    """Oracle (dense-semantics torch CPU restatement of the reference) timed on this host: reverse steps
    of ONE polblogs-shaped graph at t spread over the trajectory; graphs/s = 1 / (mean step s x 512)."""
    from oracle import edge_oracle as orc
    params = {k[len('_denoise_fn.'):]: v.detach().cpu() for k, v in model.state_dict().items()
              if k.startswith('_denoise_fn.')}
    o = orc.EdgeOracle(params, T_STEPS, int(deg.max()))
    graphs = [dict(n=n, edges=edges)]
    bt = orc.make_batch(graphs)
    bt.log_node_attr = orc.index_to_log_onehot(bt.node_attr)
    bt.log_full_edge_attr = orc.index_to_log_onehot(bt.full_edge_attr)
    g = torch.Generator().manual_seed(0)
    Nn, P = bt.num_nodes, bt.full_edge_attr.shape[0]
    ts = list(range(T_STEPS - 6, -1, -11))      # 47 steps spread over the trajectory
    times, done = [], 0
    t_start = time.perf_counter()
    for t in ts:
        tape = orc.CallbackTape(lambda tag, shape: torch.rand(shape, generator=g))
        with torch.no_grad():
            bt.log_node_attr_t, bt.log_full_edge_attr_t = orc.q_sample(o.sched, bt, torch.full((Nn,), t),
                                                                        torch.full((P,), t), tape)
            t0 = time.perf_counter()
            o.p_sample(bt, torch.full((Nn,), t), tape)
            times.append(time.perf_counter() - t0)
        done += 1
        if time.perf_counter() - t_start > budget_s and done >= 8:
            break
    step_s = float(np.mean(times))
    return {'value': 1.0 / (step_s * T_STEPS), 'unit': 'graphs/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': f'{done} reverse steps (p_sample) of 1 polblogs-shaped graph at t evenly spread over '
                      f'[0,{T_STEPS}), mean {step_s * 1e3:.0f} ms/step, extrapolated x{T_STEPS} steps'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--graphs', type=int, default=64, help='graphs sampled per GPU per step (num_generation)')
    ap.add_argument('--parts', type=int, default=0,
                    help='sub-batches (HIP streams) per sampling pass; 0 = the sampler default (3 for 64 graphs)')
    ap.add_argument('--train-steps', type=int, default=10, help='secondary metric: timed training steps (0 = skip)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    a = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    dist = world > 1
    if dist:
        import torch.distributed as td
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.cuda.set_device(local)
        td.init_process_group('nccl', device_id=torch.device('cuda', local))
    device = f'cuda:{local}'
    torch.cuda.set_device(local)

    import edgehip
    from edgehip import capi, synth
    n, edges = synth.polblogs_like(0)
    model, deg = make_model(n, edges, device)
    B = a.graphs
    gid_base, count = edgehip.shard.shard_generations(world * B, rank, world)     # weak scaling: B per rank
    assert count == B
    batch = edgehip.GraphBatch([n] * B, degree=[deg] * B, device=device, gid_base=gid_base)

    parts = a.parts if a.parts > 0 else model.default_parts(B)

    def one_pass(i, parts=parts):
        return model.sample_batch(batch, seed=1000 + i, parts=parts)

    def barrier():
        if dist:
            td.barrier()
        torch.cuda.synchronize()

    for i in range(a.warmup):
        one_pass(-1 - i)
    barrier()
    t0 = time.perf_counter()
    for i in range(a.steps):
        out = one_pass(i)
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = edgehip.shard.max_over_ranks(elapsed, device)
    edges_out = int(out._slice_dict['edge_index'][-1].item())
    value = world * B * a.steps / elapsed

    res = {'metric': 'sampled graphs/sec, polblogs 512 steps', 'value': value, 'unit': 'graphs/s', 'n_gpus': world,
           'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': 1e3 * elapsed / a.steps, 'higher_is_better': True,
           'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
           'dtype_note': 'fp32 state, weights, accumulation and results; the dense products run on the bf16 matrix pipe as '
                         'six products of exact three-way bf16 splits of both fp32 operands (logit error vs the reference '
                         'unchanged: 5e-7, DESIGN.md 5)',
           'config': {'workload': 'polblogs-shaped synthetic graph (n 1222, E 16714, d_max ~351; the pickle is never '
                                  'loaded), T=512 linear schedule, TGNN_degree_guided dim 64 heads 8/8/8/8/1 random '
                                  f'init, empirical d0, {B} generations per GPU per step', 'graphs_per_gpu': B,
                      'diffusion_steps': T_STEPS, 'nodes_per_graph': n, 'sharding': f'sample-parallel x{world}',
                      'streams_per_gpu': parts},
           'mean_edges_per_sampled_graph': edges_out / B,
           'path_hbm': {'algorithmic_GB_per_graph': SURVEY_GB_PER_GRAPH,
                        'achieved_GBps_per_gpu': value / world * SURVEY_GB_PER_GRAPH,
                        'frac_of_hbm_peak': value / world * SURVEY_GB_PER_GRAPH / HBM_PEAK_GBPS}}

    if not a.no_roofline:
        # one more pass of the same workload with HIP events around every launch (on the launch stream).  This pass
        # runs the batch as ONE block of graphs on one stream: in the timed passes blocks of graphs overlap on
        # several streams, and the duration of a launch that shares the chip with another stream's launches is
        # not a property of the kernel (`python bench.py --parts 1` times that configuration end to end)
        import ctypes
        rows_done = ctypes.c_ulonglong(0)
        torch.cuda.synchronize()
        capi.lib.edgehip_prof_layer_rows(ctypes.byref(rows_done), 1)
        capi.lib.edgehip_prof_enable(1)
        one_pass(10 ** 6, parts=1)
        torch.cuda.synchronize()
        capi.lib.edgehip_prof_layer_rows(ctypes.byref(rows_done), 1)
        prof = capi.prof_collect()
        capi.lib.edgehip_prof_enable(0)
        tot = sum(v[0] for v in prof.values())
        kern = {k: {'ms': round(v[0], 3), 'launches': v[1], 'avg_us': round(1e3 * v[0] / v[1], 3),
                    'share': round(v[0] / tot, 4)} for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])}
        dom = next(iter(kern))
        # rows per launch that were really processed: graphs without active pairs are skipped by the kernels
        Nn = rows_done.value / kern['layer']['launches'] if dom == 'layer' else n * B
        w = kernel_work(dom, Nn, kern[dom]['launches'])
        roof = None
        if w is not None:
            fl, by = w
            dur = kern[dom]['avg_us'] * 1e-6
            if fl > 0 and fl / by > MFMA_F32_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBPS * 1e9):
                ach = fl / dur / 1e12
                roof = {'kernel': dom, 'bound': 'mfma', 'achieved': ach, 'peak': MFMA_F32_PEAK_TFLOPS,
                        'unit': 'TFLOP/s', 'frac': ach / MFMA_F32_PEAK_TFLOPS, 'traffic': None}
            else:
                ach = by / dur / 1e9
                roof = {'kernel': dom, 'bound': 'hbm', 'achieved': ach, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                        'frac': ach / HBM_PEAK_GBPS, 'traffic': None}
            if roof['bound'] == 'mfma':
                # the fp32 products are EXECUTED as six bf16 MFMA products (same fp32 result to rounding): the
                # figure above prices the algorithmic fp32 flops against the fp32 matrix peak, this one the executed
                # bf16 flops against the bf16 peak
                ex = BF16_PRODUCTS_PER_FP32 * fl / dur / 1e12
                roof['executed'] = {'instruction': 'v_mfma_f32_32x32x16_bf16, 6 products per fp32 product',
                                    'achieved': ex, 'peak': MFMA_BF16_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                                    'frac': ex / MFMA_BF16_PEAK_TFLOPS}
            roof['traffic'], roof['traffic_source'] = pmc_traffic(dom)
            roof['measured'] = 'one extra pass, whole batch on one stream (kernel alone on the chip)'
            roof['avg_launch_us'] = kern[dom]['avg_us']
            roof['rows_per_launch'] = Nn
            roof['algorithmic_bytes_per_launch'] = by
            roof['algorithmic_flops_per_launch'] = fl
        res['roofline'] = roof
        res['kernels'] = kern
    if a.train_steps > 0:
        # secondary metric: training steps/sec = elbo_bpd fwd + bwd + clip + Adam on the
        # reference's polblogs batch (4 copies of the graph with independent t, README.md:70), data-parallel
        # over the ranks (one flat-bucket gradient all-reduce per step)
        TB = 4
        tb = edgehip.GraphBatch([n] * TB, degree=[deg] * TB, edges=[edges] * TB, device=device,
                                gid_base=rank * TB, degree_dtype=np.int64)
        tmodel, _ = make_model(n, edges, device)       # a fresh copy: the sampled model stays untouched
        tmodel.rng_seed = 1234 + rank
        opt = edgehip.make_optimizer(tmodel)
        tstate = {}
        torch.manual_seed(1 + rank)
        for _ in range(2):
            edgehip.train_step(tmodel, tb, opt, clip_value=1.0, state=tstate)
        barrier()
        t0 = time.perf_counter()
        for _ in range(a.train_steps):
            bpd = edgehip.train_step(tmodel, tb, opt, clip_value=1.0, state=tstate)
        barrier()
        tel = edgehip.shard.max_over_ranks(time.perf_counter() - t0, device)
        res['train'] = {'metric': 'training steps/sec, polblogs batch 4 per GPU (fwd + bwd + clip + Adam)',
                        'value': a.train_steps / tel, 'unit': 'steps/s', 'ms_per_step': 1e3 * tel / a.train_steps,
                        'graphs_per_step': TB * world, 'last_bpd': float(bpd)}
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        res['cpu_baseline'] = cpu_baseline(model, n, edges, deg)
    if rank == 0:
        print(json.dumps(res))
    if dist:
        td.destroy_process_group()


if __name__ == '__main__':
    main()
