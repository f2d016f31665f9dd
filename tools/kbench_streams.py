"""Fixed-state, multi-stream kernel bench: the fused reverse step (edgehip_sample_step) of B polblogs-shaped graphs at a
fixed noise level t, the batch cut into `parts` blocks on separate HIP streams as the sampler does, state restored
before every repetition.  Wall time per step of the whole batch; results of the step are irrelevant (timing-only
library builds can be compared).
    python tools/kbench_streams.py [--t 100] [--graphs 64] [--parts 3] [--reps 30]"""
import argparse
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'graph-generation-edge_amd'))
import torch  # noqa: E402
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--t', type=int, nargs='+', default=[100])
    ap.add_argument('--graphs', type=int, default=64)
    ap.add_argument('--parts', type=int, default=3)
    ap.add_argument('--reps', type=int, default=30)
    a = ap.parse_args()
    import edgehip
    from edgehip import capi, synth
    from edgehip.model import _state, _workspace
    n, edges = synth.polblogs_like(0)
    model, deg = bench.make_model(n, edges, 'cuda')
    B = a.graphs
    batch = edgehip.GraphBatch([n] * B, degree=[deg] * B, edges=[edges] * B, device='cuda')
    net = model._denoise_fn
    P, folded = net.c_params()
    tabs = model._tables(batch.device, int(deg.max()))
    subs = batch.split(a.parts) if a.parts > 1 else [batch]
    cur = torch.cuda.current_stream()
    streams = [cur] + [torch.cuda.Stream() for _ in subs[1:]]
    for t in a.t:
        work = []
        for b in subs:
            tg = torch.full((b.num_graphs,), t, dtype=torch.int32, device='cuda')
            a1, a2 = b.new_adj(), b.new_adj()
            rng = capi.Rng(5)
            capi.check(capi.lib.edgehip_q_sample_edges(C.byref(b.layout), b.adj.data_ptr(), tg.data_ptr(),
                                                       tabs['q_cum'].data_ptr(), tabs['q_one'].data_ptr(), None, None,
                                                       C.byref(rng), capi.ptr(tabs['lut']), a1.data_ptr(), a2.data_ptr(),
                                                       capi.stream_ptr()))
            work.append((b, a2.clone(), b.new_adj(), _state(b)['counts'], _workspace(b, P)))
        torch.cuda.synchronize()
        rng = capi.Rng(9)
        step = capi.lib.edgehip_sample_step

        def run(reps):
            for _ in range(reps):
                for (b, start, adj, counts, ws), st in zip(work, streams):
                    with torch.cuda.stream(st):
                        adj.copy_(start, non_blocking=True)
                        capi.check(step(C.byref(b.layout), C.byref(P), folded.data_ptr(), adj.data_ptr(),
                                        b.degree_i32.data_ptr(), t, tabs['post'].data_ptr(), tabs['nmax'], None, None,
                                        C.byref(rng), capi.ptr(tabs['lut']), counts.data_ptr(), ws.data_ptr(), ws.numel(),
                                        st.cuda_stream), 'sample_step')
        for st in streams[1:]:
            st.wait_stream(cur)
        run(3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(a.reps)
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / a.reps
        print(f't={t} B={B} parts={a.parts}: {1e6 * el:.1f} us per step of the batch')


if __name__ == '__main__':
    main()
