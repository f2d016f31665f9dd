"""Where does a training step spend its time?  GPU kernel time (event timer) vs wall."""
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
    from edgehip import capi, synth
    n, edges = synth.polblogs_like(0)
    model, deg = bench.make_model(n, edges, 'cuda')
    TB = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    tb = edgehip.GraphBatch([n] * TB, degree=[deg] * TB, edges=[edges] * TB, device='cuda', degree_dtype=np.int64)
    model.rng_seed = 1
    opt = edgehip.make_optimizer(model)
    state = {}
    if len(sys.argv) > 2:
        model.train_form = sys.argv[2]
    for _ in range(3):
        edgehip.train_step(model, tb, opt, clip_value=1.0, state=state)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        edgehip.train_step(model, tb, opt, clip_value=1.0, state=state)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 10
    capi.lib.edgehip_prof_enable(1)
    for _ in range(5):
        edgehip.train_step(model, tb, opt, clip_value=1.0, state=state)
    torch.cuda.synchronize()
    prof = capi.prof_collect()
    capi.lib.edgehip_prof_enable(0)
    tot = sum(v[0] for v in prof.values()) / 5
    print(f'B={TB}: wall {wall * 1e3:.2f} ms/step; edgehip kernels {tot:.2f} ms/step in {sum(v[1] for v in prof.values()) // 5} launches')
    for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0]):
        print(f'   {k:14s} {v[0] / 5:.3f} ms  x{v[1] // 5}')


if __name__ == '__main__':
    main()
