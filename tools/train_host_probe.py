"""Host time of the sections of a training step (no device sync inside the step) against the wall per step: shows
whether the step is bound by the host issuing it or by the device."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'graph-generation-edge_amd'))
import math  # noqa: E402
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402


def main():
    import edgehip
    from edgehip import synth, train, shard
    n, edges = synth.polblogs_like(0)
    model, deg = bench.make_model(n, edges, 'cuda')
    TB = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    tb = edgehip.GraphBatch([n] * TB, degree=[deg] * TB, edges=[edges] * TB, device='cuda', degree_dtype=np.int64)
    model.rng_seed = 1
    opt = edgehip.make_optimizer(model)
    state = {}
    for _ in range(5):
        edgehip.train_step(model, tb, opt, clip_value=1.0, state=state)
    torch.cuda.synchronize()
    acc = dict(prep=0.0, log_prob=0.0, backward=0.0, clip=0.0, adam=0.0)
    N = 20
    params = list(model.parameters())
    t_all = time.perf_counter()
    for _ in range(N):
        t0 = time.perf_counter()
        model.train()
        tb._train_ws_busy = False
        arena = state['arena']
        arena.zero_and_attach()
        t1 = time.perf_counter()
        with train.use_arena(arena):
            neg_vb = model.log_prob(tb)
            P = tb.__dict__.pop('_global_entries', None)
            loss = -neg_vb.sum() / (math.log(2) * P)
            t2 = time.perf_counter()
            loss.backward()
        t3 = time.perf_counter()
        shard.allreduce_flat(arena.flat)
        arena.flat.clamp_(-1.0, 1.0)
        t4 = time.perf_counter()
        opt.step()
        model._denoise_fn.weights_changed()
        t5 = time.perf_counter()
        for k, v in zip(acc, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
            acc[k] += v
    host = time.perf_counter() - t_all
    torch.cuda.synchronize()
    wall = time.perf_counter() - t_all
    print(f'B={TB}: wall {1e3 * wall / N:.2f} ms/step, host issue time {1e3 * host / N:.2f} ms/step')
    for k, v in acc.items():
        print(f'   {k:10s} {1e3 * v / N:.3f} ms')


if __name__ == '__main__':
    main()
