"""Micro-benchmark of the three trunk GEMM instantiations at the C3 shapes (GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dun_amd import kernels
from bench import timed
be = kernels.backend()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
W = 2048
g = torch.Generator(device="cuda").manual_seed(0)
H = torch.randn(B, W, device="cuda", generator=g)
Z = torch.randn(B, W, device="cuda", generator=g)
w = torch.randn(W, W, device="cuda", generator=g) * 0.02
b = torch.randn(W, device="cuda", generator=g)
out = torch.empty(B, W, device="cuda")
gw = torch.empty(W, W, device="cuda")
fl = 2.0 * B * W * W
skws = torch.empty(2 * W * W, device="cuda")
u1 = torch.randn(B, 1, device="cuda", generator=g) * 1e-3
v1 = torch.randn(1, W, device="cuda", generator=g)
# correctness of the dX form (matters when A/B-ing DUN_GEMM_RINIT / DUN_GEMM_GM builds)
_Bs = 4096
_R = torch.randn(_Bs, W, device="cuda", generator=g)
_ref = (Z[:_Bs].double() @ w.double() + _R.double() + u1[:_Bs].double() @ v1.double())
_got = be.gemm(Z[:_Bs].contiguous(), w, _Bs, W, W, a_kcontig=True, b_kcontig=False, R=_R.clone(), U=u1[:_Bs].contiguous(), V=v1)
print("dX check: max abs err %.3e (scale %.2f)" % (float((_got.double() - _ref).abs().max()), float(_ref.abs().max())))
for rep in range(2):
    ms = timed(lambda: be.gemm(H, w, B, W, W, bias=b, relu=True, colstats=True, out=out), 10)
    print("fwd  %.3f ms  %.1f TF" % (ms, fl / ms / 1e9))
    ms = timed(lambda: be.gemm(Z, w, B, W, W, a_kcontig=True, b_kcontig=False, R=out, U=u1, V=v1, out=out), 10)
    print("dX   %.3f ms  %.1f TF  (in-place residual + rank-1 head term, as in the step)" % (ms, fl / ms / 1e9))
    mean, rstd = torch.randn(W, device="cuda", generator=g), torch.rand(W, device="cuda", generator=g) + 0.5
    ms = timed(lambda: be.gemm(Z, w, B, W, W, a_kcontig=True, b_kcontig=False, R=out, U=u1, V=v1, out=out, bn=(H, mean, rstd)), 10)
    print("dX   %.3f ms  %.1f TF  (same + BatchNorm backward statistics of the block below in the epilogue)" % (ms, fl / ms / 1e9))
    wT = w.t().contiguous()
    ms = timed(lambda: be.gemm(Z, wT, B, W, W, a_kcontig=True, b_kcontig=True, R=out, U=u1, V=v1, out=out), 10)
    print("dX   %.3f ms  %.1f TF  (same, against a transposed copy of W: both operands K-contiguous)" % (ms, fl / ms / 1e9))
    G2 = torch.randn(B, W, device="cuda", generator=g) if rep == 0 else G2
    ms = timed(lambda: be.gemm(Z, wT, B, W, W, a_kcontig=True, b_kcontig=True, R=G2, U=u1, V=v1, out=out, bn=(H, mean, rstd)), 10)
    print("dX   %.3f ms  %.1f TF  (as the step runs it: transposed W, residual read from another plane, + statistics epilogue)" % (ms, fl / ms / 1e9))
    Zs = Z * (torch.rand(B, W, device="cuda", generator=g) < 0.5)
    ms = timed(lambda: be.gemm(Zs, wT, B, W, W, a_kcontig=True, b_kcontig=True, R=G2, U=u1, V=v1, out=out, bn=(H, mean, rstd)), 10)
    print("dX   %.3f ms  %.1f TF  (same, half of the A operand zero like a ReLU-masked dU)" % (ms, fl / ms / 1e9))
    del Zs
    ms = timed(lambda: be.gemm(Z, H, W, W, B, a_kcontig=False, b_kcontig=False, out=gw), 10)
    print("dW   %.3f ms  %.1f TF" % (ms, fl / ms / 1e9), flush=True)
    ms = timed(lambda: be.gemm(Z, H, W, W, B, a_kcontig=False, b_kcontig=False, out=gw, splitk_ws=skws), 10)
    print("dW   %.3f ms  %.1f TF  (split-K workspace given, as in the step)" % (ms, fl / ms / 1e9), flush=True)
    # epilogue-free variants (what the K loop alone reaches)
    ms = timed(lambda: be.gemm(H, w, B, W, W, out=out), 10)
    print("fwd (no epilogue)  %.3f ms  %.1f TF" % (ms, fl / ms / 1e9))
    ms = timed(lambda: be.gemm(Z, w, B, W, W, a_kcontig=True, b_kcontig=False, out=out), 10)
    print("dX  (no epilogue)  %.3f ms  %.1f TF" % (ms, fl / ms / 1e9), flush=True)
