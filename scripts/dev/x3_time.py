import sys, os, torch
sys.path.insert(0, os.getcwd())
from dun_amd import kernels, _lib
from bench import timed
be = kernels.backend(); lib = _lib.lib()
g = torch.Generator(device="cuda").manual_seed(0)
M, N, K = 65536, 2048, 2048
A = torch.randn(M, K, device="cuda", generator=g); B = torch.randn(N, K, device="cuda", generator=g) * 0.02
H = torch.randn(M, N, device="cuda", generator=g)
ws = torch.empty(4 * N * K, device="cuda")
fl = 2.0 * M * N * K
for mode in (0, 1):
    lib.dun_gemm_precision(mode)
    ms = timed(lambda: be.gemm(A, B, M, N, K), 10)
    print("mode %d fwd shape: %.3f ms  %.1f TF" % (mode, ms, fl / ms / 1e9))
    # dW = dU^T H: [K=batch] contraction, both operands M/N-contiguous
    ms = timed(lambda: be.gemm(A, H, K, N, M, a_kcontig=False, b_kcontig=False, splitk_ws=ws), 10)
    out = be.gemm(A[:8192], H[:8192], K, N, 8192, a_kcontig=False, b_kcontig=False, splitk_ws=ws)
    ref = A[:8192].double().t() @ H[:8192].double()
    bound = A[:8192].abs().double().t() @ H[:8192].abs().double()
    print("mode %d dW shape: %.3f ms  %.1f TF   err/bound %.3g" % (mode, ms, fl / ms / 1e9, float(((out.double() - ref).abs() / bound).max())))
lib.dun_gemm_precision(0)
