"""Micro-benchmark of the shared-head kernels at the image-DUN shapes (rows = D*B, K = 2048, O = 10)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dun_amd import kernels
from bench import timed
be = kernels.backend()
for rows, K, O in [(12 * 4096, 2048, 10), (12 * 256, 2048, 10), (51 * 65536, 2048, 1)]:
    g = torch.Generator(device="cuda").manual_seed(0)
    H = torch.randn(rows, K, device="cuda", generator=g)
    w = torch.randn(O, K, device="cuda", generator=g) * 0.02
    b = torch.randn(O, device="cuda", generator=g)
    dact = torch.randn(rows, O, device="cuda", generator=g)
    act = be.head_forward(H, w, b)
    ref = torch.addmm(b, H, w.t())
    err = float((act - ref).abs().max() / ref.abs().max())
    dw, db, dH = be.head_backward(H, w, dact, need_dH=(O > 1))
    e2 = float((dw - dact.t() @ H).abs().max() / (dact.t() @ H).abs().max())
    e3 = float((dH - dact @ w).abs().max() / (dact @ w).abs().max()) if O > 1 else 0.0
    ms = timed(lambda: be.head_forward(H, w, b), 20)
    ms2 = timed(lambda: be.head_backward(H, w, dact, need_dH=(O > 1)), 20)
    by = 4.0 * rows * K
    print("rows %8d K %d O %2d  fwd %.4f ms %.0f GB/s (err %.1e) | bwd %.4f ms %.0f GB/s (dw err %.1e dH err %.1e)"
          % (rows, K, O, ms, by / ms / 1e6, err, ms2, by * (2 if O > 1 else 1) / ms2 / 1e6, e2, e3), flush=True)
    del H, dact, dH
