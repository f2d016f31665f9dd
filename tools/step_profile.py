"""Time of every reverse step along one sampling run (64 polblogs-shaped graphs), in buckets of 32 steps."""
import ctypes as C
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
    from edgehip.model import _state, _workspace
    n, edges = synth.polblogs_like(0)
    model, deg = bench.make_model(n, edges, 'cuda')
    B = 64
    batch = edgehip.GraphBatch([n] * B, degree=[deg] * B, device='cuda')
    model.sample_batch(batch, seed=1)            # warm-up
    L = batch.layout
    P, folded = model._denoise_fn.c_params()
    ws = _workspace(batch, P)
    tabs = model._tables(batch.device, int(deg.max()))
    rng = capi.Rng(1000)
    batch.adj.zero_()
    counts = _state(batch)['counts']
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(513)]
    s = capi.stream_ptr()
    ev[0].record()
    for k, t in enumerate(reversed(range(512))):
        capi.check(capi.lib.edgehip_sample_step(C.byref(L), C.byref(P), folded.data_ptr(), batch.adj.data_ptr(),
                                                batch.degree_i32.data_ptr(), t, tabs['post'].data_ptr(), tabs['nmax'], None,
                                                None, C.byref(rng), capi.ptr(tabs['lut']), counts.data_ptr(), ws.data_ptr(),
                                                ws.numel(), s))
        ev[k + 1].record()
    torch.cuda.synchronize()
    ms = np.array([ev[k].elapsed_time(ev[k + 1]) for k in range(512)])
    print(f'total {ms.sum():.1f} ms')
    for lo in range(0, 512, 32):
        print(f't in [{511 - lo - 31:3d},{511 - lo:3d}]: {ms[lo:lo + 32].sum():7.1f} ms  ({ms[lo:lo + 32].mean() * 1e3:7.0f} us/step)')


if __name__ == '__main__':
    main()
