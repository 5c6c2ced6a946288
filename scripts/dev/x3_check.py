import sys, os, torch
sys.path.insert(0, os.getcwd())
from dun_amd import kernels, _lib
from bench import timed
be = kernels.backend(); lib = _lib.lib()
g = torch.Generator(device="cuda").manual_seed(0)
for (M, N, K) in ((4096, 512, 256), (65536, 2048, 2048)):
    A = torch.randn(M, K, device="cuda", generator=g); B = torch.randn(N, K, device="cuda", generator=g) * 0.02
    Bt = B.t().contiguous(); At = A.t().contiguous()
    ref = (A[:2048].double() @ B.double().t())
    bound = (A[:2048].abs().double() @ B.abs().double().t())
    for mode in (0, 1):
        lib.dun_gemm_precision(mode)
        for akc, bkc in ((True, True), (True, False), (False, True), (False, False)):
            out = be.gemm(A if akc else At, B if bkc else Bt, M, N, K, a_kcontig=akc, b_kcontig=bkc)
            err = float(((out[:2048].double() - ref).abs() / bound).max())
            print("M %d N %d K %d mode %d akc %d bkc %d: max |err| / sum|a||b| = %.3g" % (M, N, K, mode, akc, bkc, err))
        if M == 65536:
            fl = 2.0 * M * N * K
            ms = timed(lambda: be.gemm(A, B, M, N, K), 10)
            print("  mode %d: %.3f ms  %.1f TF" % (mode, ms, fl / ms / 1e9))
            ms = timed(lambda: be.gemm(At, Bt, M, N, K, a_kcontig=False, b_kcontig=False), 10)
            print("  mode %d (M/N-contiguous): %.3f ms  %.1f TF" % (mode, ms, fl / ms / 1e9))
lib.dun_gemm_precision(0)
