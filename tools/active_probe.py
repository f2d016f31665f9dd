"""How many (graph, step) pairs of a sampling run have active pairs at all?  (probe for per-graph early-out)"""
import os
import sys
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
    B = 64
    batch = edgehip.GraphBatch([n] * B, degree=[deg] * B, device='cuda')
    batch.adj.zero_()
    rng = capi.Rng(1000)
    Ks, As = [], []
    for t in reversed(range(512)):
        model.p_sample(batch, t, None, None, rng)
        st = batch._state
        k = (st['kptr'][1:] - st['kptr'][:-1]).cpu().numpy()
        Ks.append(k)
    Ks = np.array(Ks)            # [512 steps (t=511..0), B]
    act = Ks >= 2
    print('fraction of (graph, step) with A_b > 0:', act.mean())
    print('steps with no active graph at all:', int((~act.any(1)).sum()))
    for lo in range(0, 512, 64):
        seg = act[lo:lo + 64]
        print(f't in [{511 - lo - 63},{511 - lo}]: active graph fraction {seg.mean():.3f}, mean K {Ks[lo:lo + 64].mean():.1f}')


if __name__ == '__main__':
    main()
