"""dW = dU^T H at the per-rank batch of an N-GPU shard: time against K (exact fp32 kernels)."""
import sys, os, torch
sys.path.insert(0, os.getcwd())
from dun_amd import kernels
from bench import timed
be = kernels.backend()
g = torch.Generator(device="cuda").manual_seed(0)
W = 2048
ws = torch.empty(4 * W * W, device="cuda")
for K in (65536, 32768, 16384, 8192):
    A = torch.randn(K, W, device="cuda", generator=g); H = torch.randn(K, W, device="cuda", generator=g)
    ms = timed(lambda: be.gemm(A, H, W, W, K, a_kcontig=False, b_kcontig=False, splitk_ws=ws), 20)
    print("K %6d: %.4f ms  %.1f TF" % (K, ms, 2.0 * W * W * K / ms / 1e9))
