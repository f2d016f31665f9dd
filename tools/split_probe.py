"""Does running the generations as several independent sub-batches on separate HIP streams pay?  (graphs do not
interact; while one sub-batch sits in its latency-bound per-graph / heavy-row launches the other's fused layer
kernel fills the chip)   usage: split_probe.py [parts=2] [graphs=64]"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'graph-generation-edge_amd'))
import torch  # noqa: E402
import bench  # noqa: E402


def main():
    import edgehip
    from edgehip import synth
    parts = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    n, edges = synth.polblogs_like(0)
    model, deg = bench.make_model(n, edges, 'cuda')
    whole = edgehip.GraphBatch([n] * B, degree=[deg] * B, device='cuda', gid_base=0)

    def run_whole(seed):
        return [model.sample_batch(whole, seed=seed, parts=1)]

    def run_split(seed):
        return [model.sample_batch(whole, seed=seed, parts=parts)]

    for fn, name in ((run_whole, 'one batch'), (run_split, f'{parts} sub-batches')):
        fn(1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(2):
            out = fn(100 + i)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 2
        ne = sum(int(o._slice_dict['edge_index'][-1]) for o in out)
        print(f'{name}: {B / dt:.1f} graphs/s ({dt * 1e3:.1f} ms per pass), {ne} edges')


if __name__ == '__main__':
    main()
