import sys, os, torch
sys.path.insert(0, os.getcwd())
from dun_amd import kernels, _lib
from bench import timed
be = kernels.backend(); lib = _lib.lib()
g = torch.Generator(device="cuda").manual_seed(0)
M, N, K = 65536, 2048, 2048
A = torch.randn(M, K, device="cuda", generator=g); B = torch.randn(N, K, device="cuda", generator=g) * 0.02
lib.dun_gemm_precision(1)
fl = 2.0 * M * N * K
for _ in range(2):
    ms = timed(lambda: be.gemm(A, B, M, N, K), 10)
    print("x3: %.3f ms  %.1f TF" % (ms, fl / ms / 1e9))
