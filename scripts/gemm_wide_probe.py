"""128 x 256-tile GEMM prototype (DUN_GEMM_WIDE=1, epilogue-free problems): results against torch and time against the
128 x 128 kernel.  Run twice: DUN_GEMM_WIDE=0 and =1.  GPU only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from bench import timed
from dun_amd import kernels
bench.import_torch()
be = kernels.backend()
g = torch.Generator(device="cuda").manual_seed(0)
tag = "wide" if os.environ.get("DUN_GEMM_WIDE", "0") != "0" else "128x128"
# correctness on small whole-tile problems (all four layouts where the hook applies, split-K included)
for (M, N, K) in ((256, 512, 64), (1024, 2048, 2048), (2048, 2048, 8192)):
    for a_kc in (True, False):
        for b_kc in (True, False):
            A = torch.randn(M, K, device="cuda", generator=g)
            Bm = torch.randn(N, K, device="cuda", generator=g)
            A_st = A if a_kc else A.t().contiguous()
            B_st = Bm if b_kc else Bm.t().contiguous()
            ws = torch.empty(8 * M * N, device="cuda")
            C = be.gemm(A_st, B_st, M, N, K, a_kcontig=a_kc, b_kcontig=b_kc, splitk_ws=ws)
            ref = (A.double() @ Bm.double().t())
            err = (C.double() - ref).abs().max().item() / ref.abs().max().item()
            print("%s  M=%d N=%d K=%d a_kc=%d b_kc=%d  rel err %.2e" % (tag, M, N, K, a_kc, b_kc, err), flush=True)
            assert err < 1e-5
M, N, K = 65536, 2048, 2048
A = torch.randn(M, K, device="cuda", generator=g)
Wm = torch.randn(N, K, device="cuda", generator=g) * 0.02
out = torch.empty(M, N, device="cuda")
ms = timed(lambda: be.gemm(A, Wm, M, N, K, out=out), 10)
print("%s  fwd shape, no epilogue: %.3f ms  %.1f TF" % (tag, ms, 2.0 * M * N * K / ms / 1e9), flush=True)
H = torch.randn(M, N, device="cuda", generator=g)
gw = torch.empty(N, N, device="cuda")
skws = torch.empty(8 * N * N, device="cuda")
ms = timed(lambda: be.gemm(A, H, N, N, M, a_kcontig=False, b_kcontig=False, out=gw, splitk_ws=skws), 10)
print("%s  dW shape (split-K): %.3f ms  %.1f TF" % (tag, ms, 2.0 * M * N * K / ms / 1e9), flush=True)
for Kx in (512, 1024, 4096):
    A2 = torch.randn(M, Kx, device="cuda", generator=g)
    W2 = torch.randn(N, Kx, device="cuda", generator=g) * 0.02
    ms = timed(lambda: be.gemm(A2, W2, M, N, Kx, out=out), 10)
    print("%s  K=%d: %.3f ms  %.1f TF" % (tag, Kx, ms, 2.0 * M * N * Kx / ms / 1e9), flush=True)
    del A2, W2
