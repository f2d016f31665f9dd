"""Kernel micro-benchmark: one reverse step (p_sample) of B polblogs-shaped graphs at a fixed noise
level t, repeated; prints per-kernel-class average launch times from the event timer.

    python tools/kbench.py [--t 100] [--graphs 64] [--reps 20]
"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'graph-generation-edge_amd'))
import torch  # noqa: E402
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--t', type=int, default=100)
    ap.add_argument('--graphs', type=int, default=64)
    ap.add_argument('--reps', type=int, default=20)
    a = ap.parse_args()
    import edgehip
    from edgehip import capi, synth
    from edgehip.model import _state
    n, edges = synth.polblogs_like(0)
    model, deg = bench.make_model(n, edges, 'cuda')
    B = a.graphs
    batch = edgehip.GraphBatch([n] * B, degree=[deg] * B, edges=[edges] * B, device='cuda')
    # x_t ~ q(.|x0) on the device
    tabs = model._tables(batch.device, int(deg.max()))
    tg = torch.full((B,), a.t, dtype=torch.int32, device='cuda')
    a1, a2 = batch.new_adj(), batch.new_adj()
    rng = capi.Rng(5)
    capi.check(capi.lib.edgehip_q_sample_edges(C.byref(batch.layout), batch.adj.data_ptr(), tg.data_ptr(),
                                               tabs['q_cum'].data_ptr(), tabs['q_one'].data_ptr(), None, None,
                                               C.byref(rng), capi.ptr(tabs['lut']), a1.data_ptr(), a2.data_ptr(),
                                               capi.stream_ptr()))
    start = a2.clone()
    for rep in range(a.reps + 3):
        if rep == 3:
            torch.cuda.synchronize()
            if hasattr(capi.lib, 'edgehip_debug_read'):
                capi.lib.edgehip_debug_read((C.c_ulonglong * 16)(), 1)
            capi.lib.edgehip_prof_enable(1)
        batch.adj.copy_(start)
        model.p_sample(batch, a.t, None, None, capi.Rng(100 + rep))
    torch.cuda.synchronize()
    prof = capi.prof_collect()
    capi.lib.edgehip_prof_enable(0)
    st = _state(batch)
    E = int(start.view(torch.uint8).to(torch.int32).sum()) if False else None
    _, sl = batch.extract_edges(start)
    print(f't={a.t} B={B} E_t/graph={int(sl[-1]) / B:.0f} K/graph={int(st["kptr"][-1]) / B:.1f} '
          f'A/graph={int(st["aptr"][-1]) / B:.0f}')
    tot = 0.0
    for k, (ms, cnt) in sorted(prof.items(), key=lambda kv: -kv[1][0]):
        per_step = ms / a.reps
        tot += per_step
        print(f'  {k:14s} {1e3 * ms / cnt:9.1f} us/launch  x{cnt // a.reps:2d}  = {1e3 * per_step:9.1f} us/step')
    print(f'  total {1e3 * tot:.1f} us/step')
    if hasattr(capi.lib, 'edgehip_debug_read'):
        buf = (C.c_ulonglong * 16)()
        capi.lib.edgehip_debug_read(buf, 1)
        tot_c = sum(buf[:9]) or 1
        names = ['0 h planes', 'A qt', 'B stream walk (decode + visits)', 'B long rows', 'B zero + barrier wait', 'C conv products',
                 'C reduce + silu', 'D gru', 'E ctx']
        print('  k_layer phase shares (thread 0 of every tile, s_memtime):', ', '.join(f'{n} {100 * buf[i] / tot_c:.1f}%' for i, n in enumerate(names)))


if __name__ == '__main__':
    main()
