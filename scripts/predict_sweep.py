"""Depth-marginal predictor kernels: algorithmic GB/s (read 4*D*n*(C|1), write 4*n*(C|1) [x2 with std])."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dun_amd import kernels
from bench import timed
be = kernels.backend()
for D, B in ((51, 1 << 16), (51, 1 << 22)):
    act = torch.randn(D, B, 1, device="cuda"); q = torch.softmax(torch.randn(D, device="cuda"), 0)
    y = torch.randn(B, 1, device="cuda"); ls = torch.zeros(1, device="cuda")
    ms = timed(lambda: be.predict(act, q, True, get_std=True, y=y, log_std=ls), 20)
    by = 4.0 * D * B + 12.0 * B
    print("gauss   D=%d B=2^%d  %.4f ms  %.0f GB/s" % (D, B.bit_length() - 1, ms, by / ms / 1e6), flush=True)
for C in (10, 100):
    for B in (4096, 1 << 20):
        D = 12
        act = torch.randn(D, B, C, device="cuda"); q = torch.softmax(torch.randn(D, device="cuda"), 0)
        y = torch.randint(0, C, (B,), device="cuda")
        ms = timed(lambda: be.predict(act, q, False, y=y), 20)
        by = 4.0 * D * B * C + 4.0 * B * C + 8.0 * B
        print("softmax D=%d C=%d B=%d  %.4f ms  %.0f GB/s" % (D, C, B, ms, by / ms / 1e6), flush=True)
