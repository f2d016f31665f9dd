"""edgehip_linear_fwd (the training GEMM: fp32 in/out, split-bf16 products) against torch.nn.functional.linear (fp32,
rocBLAS / hipBLASLt) on training shapes: time per call and effective bandwidth."""
import ctypes as C
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'graph-generation-edge_amd'))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402


def main():
    from edgehip import capi
    for (M, K, N, act) in [(4888, 128, 256, 1), (4888, 512, 128, 0), (39104, 128, 256, 1), (39104, 512, 128, 0),
                           (117312, 128, 256, 1), (117312, 512, 128, 0), (117312, 128, 512, 0), (117312, 64, 64, 0)]:
        x = torch.randn(M, K, device='cuda')
        w = torch.randn(N, K, device='cuda') / K ** 0.5
        b = torch.randn(N, device='cuda')
        y = torch.empty(M, N, device='cuda')
        s = capi.stream_ptr()

        def mine():
            capi.check(capi.lib.edgehip_linear_fwd(x.data_ptr(), M, K, w.data_ptr(), b.data_ptr(), N, act, y.data_ptr(), s))

        def ref():
            r = F.linear(x, w, b)
            return F.silu(r) if act else r
        res = []
        for f in (mine, ref):
            for _ in range(3):
                f()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                f()
            torch.cuda.synchronize()
            res.append((time.perf_counter() - t0) / 20 * 1e6)
        byt = 4.0 * (M * K + M * N + N * K)
        err = float((y - ref()).abs().max())
        print(f'M {M:6d} K {K:3d} N {N:3d} act {act}: edgehip {res[0]:7.1f} us ({byt / res[0] / 1e6:5.2f} TB/s)   torch {res[1]:7.1f} us   max|diff| {err:.1e}')


if __name__ == '__main__':
    main()
