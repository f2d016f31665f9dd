"""bench.py - sampled graphs/sec of the degree-guided reverse process, polblogs shape, T = 512.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python bench.py --gpus 8 ...                      (starts 8 ranks itself when not under a launcher)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one batch: model.sample() of the rank's graphs through all 512 reverse
steps, state resident in HBM, ending with the edge lists on the device.  Graphs of a sampling run are independent:
they are sharded over the ranks by global graph id with no data-path collective (the RNG is keyed by that id, so
the sampled graphs do not depend on the sharding).
  weak scaling (default):   every rank samples `--graphs` (64) graphs per step;
  strong scaling:           `--total-graphs 64` = BASELINE config 3, the 64 generations split 64/N per rank.
Under N > 1 the weak run also times the strong configuration and reports it as `strong_scaling`.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'graph-generation-edge_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

T_STEPS = 512
HEADS = [8, 8, 8, 8, 1]
MFMA_F32_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md, Peak FP32 (matrix)
MFMA_BF16_PEAK_TFLOPS = 2500.0    # MI355X_MICROARCH.md, BF16 dense
BF16_PRODUCTS_PER_FP32 = 6        # csrc/gemm.h: exact 3-way bf16 split of both operands, six products kept
HBM_PEAK_GBPS = 8000.0            # MI355X_MICROARCH.md, HBM3E peak
SURVEY_GB_PER_GRAPH = 40.7        # bytes of the UNFOLDED formulation per sampled polblogs graph
PMC_FILES = ('r03_pmc_traffic.json', 'r02_pmc_traffic.json', 'r01_pmc_traffic.json')     # newest first


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--graphs', type=int, default=64, help='weak scaling: graphs sampled per GPU per step (num_generation)')
    ap.add_argument('--total-graphs', type=int, default=0,
                    help='strong scaling: graphs per step over ALL ranks (64 = BASELINE config 3); overrides --graphs')
    ap.add_argument('--parts', type=int, default=0,
                    help='sub-batches (HIP streams) per sampling pass; 0 = the sampler default (3 for 64 graphs)')
    ap.add_argument('--train-steps', type=int, default=10, help='secondary metric: timed training steps (0 = skip)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--rehearse-on-one-gpu', action='store_true',
                    help='NOT a measurement: all ranks share cuda:0 and talk over gloo, to exercise the multi-rank code '
                         'paths (launcher, sharding, strong-scaling leg, data-parallel training) on a one-GPU box')
    ap.add_argument('--dry-run-cpu', action='store_true',
                    help='NOT a measurement: no GPU work at all; the ranks meet over gloo and print the multi-rank '
                         'bookkeeping of the JSON line (CPU test of the launcher)')
    ap.add_argument('--fail-rank', type=int, default=-1, help='with --dry-run-cpu: this rank exits with code 7')
    return ap.parse_args()


def launch_ranks(n, argv=None, timeout_s=3000.0):
    """`python bench.py --gpus N` outside a launcher: start N ranks of this script as CHILD processes (one per GPU,
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment) and relay rank 0's JSON line.  This process has
    not touched the GPU and never does (no exec from a process that initialised it).  All children are polled: when
    one exits non-zero (bad device, import error) the others - which would sit in the rendezvous or a collective until
    their own timeout - are killed and that return code is reported; a rendezvous port lost to another process between
    the probe and rank 0's bind gets one retry on a fresh port."""
    argv = sys.argv[1:] if argv is None else list(argv)

    def attempt():
        with socket.socket() as s:
            s.bind(('127.0.0.1', 0))
            port = s.getsockname()[1]
        procs = []
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1',
                       MASTER_PORT=str(port), LOCAL_WORLD_SIZE=str(n))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                          stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL,
                                          stderr=subprocess.PIPE if r == 0 else None))
        # rank 0's pipes are drained by threads so that polling never blocks on them
        import threading
        bufs = {'out': b'', 'err': b''}

        def drain(name, fh):
            bufs[name] = fh.read()
        readers = [threading.Thread(target=drain, args=('out', procs[0].stdout), daemon=True),
                   threading.Thread(target=drain, args=('err', procs[0].stderr), daemon=True)]
        for t in readers:
            t.start()
        deadline, rc, failed = time.time() + timeout_s, 0, None
        while True:
            codes = [p.poll() for p in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad and failed is None:
                failed, rc = bad[0][0], bad[0][1]
                grace = time.time() + 5.0                     # the others usually follow by themselves
                while time.time() < grace and any(p.poll() is None for p in procs):
                    time.sleep(0.05)
                for p in procs:
                    if p.poll() is None:
                        p.kill()                              # these exact children
            if all(c is not None for c in [p.poll() for p in procs]):
                break
            if time.time() > deadline:
                for p in procs:
                    if p.poll() is None:
                        p.kill()
                rc = rc or 124
            time.sleep(0.05)
        for t in readers:
            t.join(timeout=5.0)
        if failed is not None:
            sys.stderr.write(f'bench.py: rank {failed} exited with code {rc}; the remaining ranks were stopped\n')
        return rc, bufs['out'].decode(errors='replace'), bufs['err'].decode(errors='replace')

    rc, out, err = attempt()
    if rc and ('EADDRINUSE' in err or 'address already in use' in err.lower()):
        rc, out, err = attempt()
    sys.stderr.write(err)
    # stdout carries the ONE JSON line; whatever else a library printed there (gloo's connection notice in a rehearsal)
    # goes to stderr
    for line in out.splitlines():
        (sys.stdout if line.lstrip().startswith('{"metric"') else sys.stderr).write(line + '\n')
    sys.stdout.flush()
    return rc


def certify_ranks(td, torch, device, elapsed_local, dry=False):
    """What lets a reader of the JSON line confirm the multi-GPU run was one: the world size the process group reports,
    every rank's (rank, host, pid, device index, device name / uuid / PCI bus id) gathered over the group itself, the
    RCCL version torch was built against, and every rank's own elapsed time (the line's value uses the MAX)."""
    me = {'rank': td.get_rank(), 'host': socket.gethostname(), 'pid': os.getpid(), 'elapsed_s': elapsed_local}
    if not dry:
        idx = torch.cuda.current_device()
        pr = torch.cuda.get_device_properties(idx)
        me.update({'device': idx, 'name': pr.name, 'uuid': str(getattr(pr, 'uuid', '')),
                   'pci': '%04x:%02x:%02x' % (getattr(pr, 'pci_domain_id', 0), getattr(pr, 'pci_bus_id', 0),
                                              getattr(pr, 'pci_device_id', 0)),
                   'visible': os.environ.get('HIP_VISIBLE_DEVICES') or os.environ.get('ROCR_VISIBLE_DEVICES') or 'all'})
    got = [None] * td.get_world_size()
    td.all_gather_object(got, me)
    out = {'world_size': td.get_world_size(), 'backend': td.get_backend(), 'ranks': got,
           'per_rank_elapsed_s': [g['elapsed_s'] for g in got]}
    if not dry:
        try:
            out['rccl_version'] = '.'.join(str(x) for x in torch.cuda.nccl.version())
        except Exception as e:                                 # noqa: BLE001 - a missing version string is not fatal
            out['rccl_version'] = f'unavailable ({type(e).__name__})'
        out['distinct_devices'] = len({(g['host'], g.get('uuid') or g.get('pci'), g.get('device')) for g in got})
    return out


def dry_run_cpu(a):
    """`--dry-run-cpu` (tests/test_bench_launcher.py): NOT a measurement and not the hot path - the ranks meet over gloo on
    the CPU, skip the sampling (there is no GPU to run it on) and print the line's multi-rank bookkeeping, so the
    launcher, the environment plumbing and the certificate can be checked without a GPU."""
    import torch
    import torch.distributed as td
    from edgehip import shard
    world, rank = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0'))
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    td.init_process_group('gloo')
    if a.fail_rank == rank:
        sys.exit(7)
    total = a.total_graphs if a.total_graphs > 0 else world * a.graphs
    gid_base, count = shard.shard_generations(total, rank, world)
    td.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))                             # stand-in for the rank's passes
    td.barrier()
    local = time.perf_counter() - t0
    el = shard.max_over_ranks(local, 'cpu')
    shards = [None] * world
    td.all_gather_object(shards, (gid_base, count))
    res = {'metric': 'sampled graphs/sec, polblogs 512 steps', 'value': None, 'unit': 'graphs/s', 'n_gpus': world,
           'steps': a.steps, 'warmup': a.warmup, 'dry_run': 'no GPU work was done: launcher / process-group bookkeeping only',
           'scaling': 'strong' if a.total_graphs > 0 else 'weak', 'elapsed_max_s': el,
           'config': {'total_graphs_per_step': total, 'shards': shards},
           'multi_gpu': certify_ranks(td, torch, 'cpu', local, dry=True)}
    if rank == 0:
        print(json.dumps(res))
    td.destroy_process_group()


def make_model(n, edges, device, T=T_STEPS):
    from types import SimpleNamespace
    import torch
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


def layer_work(rows, head_visits=0.0):
    """Algorithmic flops / HBM bytes of ONE launch of the fused layer kernel over `rows` node rows (DESIGN.md
    'Kernels'), averaged over the 5 layers.  Flops are those of the FOLDED formulation the library executes (keys /
    values never formed): q~ = h Mcat, conv = [z|h] Ocat, GRU (6 x 64x64), ctx 64->256 - the dense products - plus
    the attention of ALL rows, which the kernel walks itself: per (target, neighbour, head) visit a 64-channel score
    and a 64-channel weighted sum = 256 fp32 flops on the vector ALU (`head_visits` per launch, counted by the kernel).
    Returns (all flops, bytes, dense-product flops)."""
    import numpy as np
    dense = np.mean([2.0 * rows * (64 * 64 * h + (64 * h + 64) * 64 + 6 * 64 * 64 + 64 * 256) for h in HEADS])
    fl = dense + 256.0 * head_visits
    # HBM: read h row + bit row (~160 B), write h' row + per-tile context sums; q~/z stay in LDS, neighbour rows
    # and weights are L2-resident re-reads
    by = rows * (256 + 160 + 256) + (rows / 32) * 1024
    return fl, by, dense


def train_flops(Nn, directed_edges, A):
    """Algorithmic flops of one training step (forward of the UNFOLDED network as the reference runs it, layers.py:
    315-383, + backward = 2x forward: dX and dW products)."""
    node = 2 * 192 * 64 + 2 * (256 * 128 + 128 * 128 + 128 * 128)
    for h in HEADS:
        node += 3 * 2 * 128 * 64 * h + 2 * 128 * 64 + 2 * 2 * 64 * 192 + 2 * 128 * 256 + 2 * 256 * 64
    edge = sum(256 * h for h in HEADS)                # per directed edge and head: score (2*64) + weighted sum (2*64)
    pair = 2 * (128 * 128 + 128 * 64 + 64 * 2)
    return 3.0 * (Nn * node + directed_edges * edge + A * pair)


def pmc_traffic():
    """Per-kernel HBM bytes per launch from the committed PMC passes (rocprofv3 cannot run inside this process):
    profiles/r0X_pmc_traffic.json, made by tools/pmc_traffic.py from two `rocprofv3 --pmc` runs (FETCH_SIZE,
    WRITE_SIZE) of `bench.py --parts 1 --steps 1 --warmup 0`; None when absent."""
    for name in PMC_FILES:
        path = os.path.join(ROOT, 'profiles', name)
        if os.path.exists(path):
            with open(path) as fh:
                return json.load(fh)['kernels'], 'profiles/' + name
    return None, None


def cpu_baseline(model, n, edges, deg, budget_s=30.0, B=4):
    """Oracle (dense-semantics torch CPU restatement of the reference) timed on this host, as SURVEY 8d plans it:
    reverse steps of a batch of B = 4 polblogs-shaped graphs (the reference's polblogs batch) at t spread over the
    trajectory; graphs/s = B / (mean step s x 512).  Bounded: 47 steps unless the budget runs out first (>= 8)."""
    import numpy as np
    import torch
    from oracle import edge_oracle as orc
    params = {k[len('_denoise_fn.'):]: v.detach().cpu() for k, v in model.state_dict().items()
              if k.startswith('_denoise_fn.')}
    o = orc.EdgeOracle(params, T_STEPS, int(deg.max()))
    graphs = [dict(n=n, edges=edges)] * B
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
    return {'value': B / (step_s * T_STEPS), 'unit': 'graphs/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': f'{done} reverse steps (p_sample) of a batch of B = {B} polblogs-shaped graphs at t evenly spread '
                      f'over [0,{T_STEPS}) (states drawn by forward noising of the true graph), mean '
                      f'{step_s * 1e3:.0f} ms per step of the batch; graphs/s = B / (step x {T_STEPS}): an extrapolation '
                      f'from {done} of the {T_STEPS} steps, not a whole trajectory'}


def main():
    a = parse_args()
    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(launch_ranks(a.gpus))
    if a.dry_run_cpu:
        return dry_run_cpu(a)

    import numpy as np
    import torch
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    dist = world > 1
    if a.rehearse_on_one_gpu:
        local = 0
        import faulthandler
        faulthandler.dump_traceback_later(75, exit=True)      # a rehearsal that hangs says where
    if dist:
        import torch.distributed as td
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.cuda.set_device(local)
        if a.rehearse_on_one_gpu:
            td.init_process_group('gloo')
        else:
            td.init_process_group('nccl', device_id=torch.device('cuda', local))
    device = f'cuda:{local}'
    torch.cuda.set_device(local)

    import edgehip
    from edgehip import capi, synth
    n, edges = synth.polblogs_like(0)
    model, deg = make_model(n, edges, device)

    def barrier():
        if dist:
            td.barrier()
        torch.cuda.synchronize()

    def timed(total, steps, warmup, parts_arg, seed0):
        """`steps` passes of `total` graphs over all ranks -> (elapsed max over ranks, this rank's batch, parts, last out)"""
        gid_base, count = edgehip.shard.shard_generations(total, rank, world)
        batch = edgehip.GraphBatch([n] * count, degree=[deg] * count, device=device, gid_base=gid_base)
        parts = parts_arg if parts_arg > 0 else model.default_parts(count)
        out = None
        for i in range(warmup):
            out = model.sample_batch(batch, seed=seed0 - 1 - i, parts=parts) if count else None
        barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            out = model.sample_batch(batch, seed=seed0 + i, parts=parts) if count else None
        barrier()
        mine = time.perf_counter() - t0
        el = edgehip.shard.max_over_ranks(mine, device)
        return el, batch, parts, out, mine

    strong = a.total_graphs > 0
    total = a.total_graphs if strong else world * a.graphs
    elapsed, batch, parts, out, elapsed_local = timed(total, a.steps, a.warmup, a.parts, 1000)
    B = batch.num_graphs
    edges_out = int(out._slice_dict['edge_index'][-1].item()) if out is not None else 0
    value = total * a.steps / elapsed

    res = {'metric': 'sampled graphs/sec, polblogs 512 steps', 'value': value, 'unit': 'graphs/s', 'n_gpus': world,
           'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': 1e3 * elapsed / a.steps, 'higher_is_better': True,
           'scaling': 'strong' if strong else 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
           'dtype_note': 'fp32 state, weights, accumulation and results; the dense products run on the bf16 matrix pipe as '
                         'six products of exact three-way bf16 splits of both fp32 operands (logit error vs the reference '
                         'unchanged: 5e-7, DESIGN.md 5)',
           'config': {'workload': 'polblogs-shaped synthetic graph (n 1222, E 16714, d_max ~351; the pickle is never '
                                  'loaded), T=512 linear schedule, TGNN_degree_guided dim 64 heads 8/8/8/8/1 random '
                                  f'init, empirical d0, {total} generations per step over {world} GPU(s)',
                      'graphs_per_gpu': B, 'total_graphs_per_step': total, 'diffusion_steps': T_STEPS,
                      'nodes_per_graph': n, 'sharding': f'sample-parallel x{world}', 'streams_per_gpu': parts},
           'mean_edges_per_sampled_graph': edges_out / max(B, 1)}
    if a.rehearse_on_one_gpu:
        res['rehearsal'] = 'all ranks on cuda:0 over gloo: exercises the multi-rank code paths, the numbers mean nothing'

    if dist:
        # the run certifies itself: world size, devices and per-rank times as the process group saw them
        res['multi_gpu'] = certify_ranks(td, torch, device, elapsed_local)
    if dist and not strong:
        # BASELINE config 3 as written: the 64 generations of ONE sampling run split over the ranks (64/N each)
        ssteps = max(1, min(a.steps, 3))
        sel, sb, sparts, _, smine = timed(64, ssteps, 1, a.parts, 5000)
        per_rank = [None] * world
        td.all_gather_object(per_rank, smine)
        res['strong_scaling'] = {'total_graphs_per_step': 64, 'graphs_per_gpu': sb.num_graphs, 'streams_per_gpu': sparts,
                                 'value': 64 * ssteps / sel, 'unit': 'graphs/s', 'steps': ssteps,
                                 'ms_per_step': 1e3 * sel / ssteps, 'per_rank_elapsed_s': per_rank}
    elif dist:
        # --total-graphs: the line itself IS the strong-scaling point; the block repeats it in the same shape
        res['strong_scaling'] = {'total_graphs_per_step': total, 'graphs_per_gpu': B, 'streams_per_gpu': parts,
                                 'value': value, 'unit': 'graphs/s', 'steps': a.steps,
                                 'ms_per_step': 1e3 * elapsed / a.steps,
                                 'per_rank_elapsed_s': res['multi_gpu']['per_rank_elapsed_s']}

    if not a.no_roofline and B > 0 and rank == 0:
        # One more pass of the same workload with HIP events around every launch (on the launch stream) and the
        # per-step work counters.  It runs the batch as ONE block of graphs on one stream: in the timed passes blocks
        # of graphs overlap on several streams, and the duration of a launch that shares the chip with another
        # stream's launches is not a property of the kernel (`python bench.py --parts 1` times that end to end).
        import ctypes
        rows_done, visits_done = ctypes.c_ulonglong(0), ctypes.c_ulonglong(0)
        stats = torch.zeros((T_STEPS, 8), dtype=torch.int64, device=device)
        torch.cuda.synchronize()
        capi.lib.edgehip_prof_enable(1)
        capi.lib.edgehip_prof_layer_rows(ctypes.byref(rows_done), 1)
        capi.lib.edgehip_prof_layer_visits(ctypes.byref(visits_done), 1)
        t0 = time.perf_counter()
        model.sample_batch(batch, seed=1000 + a.steps - 1, parts=1, step_stats=stats)    # the last timed pass again
        torch.cuda.synchronize()
        one_stream_s = time.perf_counter() - t0
        capi.lib.edgehip_prof_layer_rows(ctypes.byref(rows_done), 1)
        capi.lib.edgehip_prof_layer_visits(ctypes.byref(visits_done), 1)
        prof = capi.prof_collect()
        capi.lib.edgehip_prof_enable(0)
        tot = sum(v[0] for v in prof.values())
        kern = {k: {'ms': round(v[0], 3), 'launches': v[1], 'avg_us': round(1e3 * v[0] / v[1], 3),
                    'share': round(v[0] / tot, 4)} for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])}
        dom = next(iter(kern))
        roof = None
        if dom == 'layer':
            rows = rows_done.value / kern['layer']['launches']      # rows per launch really processed
            hvis = visits_done.value / kern['layer']['launches']    # (target, neighbour, head) visits per launch
            fl, by, dense = layer_work(rows, hvis)
            dur = kern[dom]['avg_us'] * 1e-6
            ach = fl / dur / 1e12
            ex = BF16_PRODUCTS_PER_FP32 * dense / dur / 1e12
            roof = {'kernel': 'k_layer<h> (fused GNN layer: dense products on the matrix pipe + the attention of every row '
                              'on the vector ALU)', 'bound': 'mfma', 'achieved': ach, 'peak': MFMA_F32_PEAK_TFLOPS,
                    'unit': 'TFLOP/s', 'frac': ach / MFMA_F32_PEAK_TFLOPS, 'traffic': None,
                    'peak_note': 'fp32 peak of the chip, 157.3 TFLOP/s on the matrix pipe and the same on the vector ALU: the '
                                 'kernel\'s algorithmic fp32 flops (dense products + attention) are priced against it',
                    'dense_products_only': {'achieved': dense / dur / 1e12, 'frac': dense / dur / 1e12 / MFMA_F32_PEAK_TFLOPS,
                                            'note': 'the figure of rounds 1-3 (attention not counted); until round 3 the rows '
                                                    'above 48 neighbours were walked by a kernel of their own'},
                    # the fp32 products are EXECUTED as six bf16 MFMA products: the figures above price algorithmic fp32
                    # flops against the fp32 peak, this one the executed bf16 flops against the bf16 matrix peak
                    'executed': {'instruction': 'v_mfma_f32_32x32x16_bf16, 6 products per fp32 product', 'achieved': ex,
                                 'peak': MFMA_BF16_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': ex / MFMA_BF16_PEAK_TFLOPS},
                    'measured': 'one extra pass, whole batch on one stream (kernel alone on the chip)',
                    'avg_launch_us': kern[dom]['avg_us'], 'rows_per_launch': rows, 'head_visits_per_launch': hvis,
                    'algorithmic_bytes_per_launch': by, 'algorithmic_flops_per_launch': fl,
                    'dense_flops_per_launch': dense, 'attention_flops_per_launch': fl - dense,
                    'hbm_frac_of_this_kernel': by / dur / 1e9 / HBM_PEAK_GBPS}
        res['roofline'] = roof
        res['kernels'] = kern

        # ---- the work of the timed trajectory (it depends on how fast THIS network adds edges)
        st = stats.cpu().numpy()[::-1]              # row 0 = first reverse step (t = T-1)
        nodes_all = n * B
        idx = list(range(0, T_STEPS, 32)) + [T_STEPS - 1]
        res['trajectory'] = {
            'note': 'per reverse step, summed over the batch; every 32nd step listed (first = t 511). idle graph = '
                    'fewer than 2 active nodes (skipped); live rows = node rows left after isolated-class compaction',
            't': [T_STEPS - 1 - i for i in idx],
            'edges_E_t': [int(st[i, 6] // 2) for i in idx],
            'active_nodes_K': [int(st[i, 0]) for i in idx],
            'active_pairs_A': [int(st[i, 1]) for i in idx],
            'non_idle_graphs': [int(st[i, 2]) for i in idx],
            'live_rows': [int(st[i, 3]) for i in idx],
            'rows_above_48_neighbours': [int(st[i, 4]) for i in idx],
            'total_pair_decisions': int(st[:, 1].sum()),
            'idle_graph_step_fraction': float(1.0 - st[:, 2].sum() / (B * T_STEPS)),
            'live_row_fraction_of_all_rows': float(st[:, 3].sum() / (nodes_all * T_STEPS)),
            'live_row_fraction_of_non_idle_rows': float(st[:, 3].sum() / max(st[:, 7].sum(), 1))}

        # ---- achieved HBM GB/s of the whole path, from the counters
        kk, src = pmc_traffic()
        hbm = {'unfolded_formulation_equivalent': {
            'note': 'NOT a measurement: graphs/s x the 40.7 GB per graph the UNFOLDED reference formulation would move '
                    '(SURVEY.md 8d); the kernels here never move those bytes',
            'GBps_per_gpu': value / world * SURVEY_GB_PER_GRAPH}}
        if kk is not None and B == 64:
            per_pass = sum(v['hbm_bytes_per_launch'] * v['launches'] for v in kk.values())
            hbm.update({'source': src + ' (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this script at --parts 1, '
                                        'gfx950 corrections of MI355X_MICROARCH.md)',
                        'counter_bytes_per_pass': per_pass, 'GB_per_graph': per_pass / 64 / 1e9,
                        'achieved_GBps': per_pass / (elapsed / a.steps) / 1e9,
                        'frac_of_hbm_peak': per_pass / (elapsed / a.steps) / 1e9 / HBM_PEAK_GBPS})
            if roof is not None:
                sel = [v for k, v in kk.items() if k.startswith('k_layer')]
                if sel:
                    nl = sum(v['launches'] for v in sel)
                    roof['traffic'] = sum(v['hbm_bytes_per_launch'] * v['launches'] for v in sel) / nl
                    roof['traffic_source'] = src
        res['path_hbm'] = hbm
        res['one_stream_pass_ms'] = 1e3 * one_stream_s

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
        # roofline of the step: algorithmic flops of fwd + bwd at the mean (E_t, A) of 16 more (untimed) steps with
        # the same t distribution, against the fp32 matrix peak (the products run as exact-split bf16 MFMAs, priced as fp32 work)
        As, Es = [], []
        for _ in range(16):
            edgehip.train_step(tmodel, tb, opt, clip_value=1.0, state=tstate)
            As.append(int(tb.active_row.numel()))
            Es.append(tb._state['deg_t'][:tb.num_nodes].sum())
        torch.cuda.synchronize()
        dirE = float(np.mean([int(e) for e in Es]))
        fl = train_flops(tb.num_nodes, dirE, float(np.mean(As)))
        ach = fl / (tel / a.train_steps) / 1e12
        res['train']['roofline'] = {'bound': 'mfma', 'achieved': ach, 'peak': MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                                    'frac': ach / MFMA_F32_PEAK_TFLOPS, 'algorithmic_flops_per_step': fl,
                                    'mean_active_pairs': float(np.mean(As)), 'mean_directed_edges': dirE,
                                    'note': 'whole step (~240 launches, latency-bound at 4 888 node rows), not one kernel'}
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        res['cpu_baseline'] = cpu_baseline(model, n, edges, deg)
    if rank == 0:
        print(json.dumps(res))
    if dist:
        td.destroy_process_group()


if __name__ == '__main__':
    main()
