"""Where do the non-MFMA cycles of the trunk GEMM go?  Time of the epilogue-free kernel against K at the C3 tile count
(M=65536, N=2048: 8192 tiles): the intercept of time(K) is the per-tile fixed cost (prologue fill, epilogue stores, wave
launch), the slope the K loop itself.  GPU only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from bench import timed
from dun_amd import kernels
bench.import_torch()
be = kernels.backend()
M, N = 65536, 2048
g = torch.Generator(device="cuda").manual_seed(0)
out = torch.empty(M, N, device="cuda")
res = []
for K in (256, 512, 1024, 2048, 4096):
    A = torch.randn(M, K, device="cuda", generator=g)
    Bm = torch.randn(N, K, device="cuda", generator=g) * 0.02
    ms = timed(lambda: be.gemm(A, Bm, M, N, K, out=out), 10)
    res.append((K, ms))
    print("K=%5d  %.3f ms  %.1f TF  (%.3f of 157.3)" % (K, ms, 2.0 * M * N * K / ms / 1e9, 2.0 * M * N * K / ms / 1e9 / 157.3), flush=True)
    del A, Bm
(k0, t0), (k1, t1) = res[-2], res[-1]
slope = (t1 - t0) / (k1 - k0)
print("slope %.4f us per K step of 1 -> loop alone runs at %.1f TF (%.3f); intercept %.3f ms = %.1f %% of the K=2048 launch" % (
    slope * 1e3, 2.0 * M * N / slope / 1e9, 2.0 * M * N / slope / 1e9 / 157.3, t0 - slope * k0, 100 * (t0 - slope * k0) / t0))
