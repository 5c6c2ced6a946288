"""Probe: does a memory-bound elementwise kernel hide under a concurrently running trunk GEMM (second stream)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dun_amd import kernels
be = kernels.backend()
B, W = 65536, 2048
g = torch.Generator(device="cuda").manual_seed(0)
H = torch.randn(B, W, device="cuda", generator=g); Z = torch.randn(B, W, device="cuda", generator=g)
gw = torch.empty(W, W, device="cuda"); ws = torch.empty(2 * W * W, device="cuda")
a = torch.randn(B, W, device="cuda", generator=g); b = torch.randn(B, W, device="cuda", generator=g); c = torch.empty_like(a)
s2 = torch.cuda.Stream()
def gemm(): be.gemm(Z, H, W, W, B, a_kcontig=False, b_kcontig=False, out=gw, splitk_ws=ws)
def mem():
    torch.add(a, b, out=c); torch.mul(a, b, out=c)   # 2 x 1.5 GB of traffic, like bn_bwd_reduce + bn_bwd_apply
def run(n, mode):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        if mode == "seq":
            gemm(); mem()
        else:
            s2.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s2):
                mem()
            gemm()
            torch.cuda.current_stream().wait_stream(s2)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for mode in ("seq", "par", "seq", "par"):
    run(3, mode); print(mode, "%.3f ms per (GEMM + 3 GB elementwise)" % run(20, mode), flush=True)
