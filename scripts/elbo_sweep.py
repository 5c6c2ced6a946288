"""Size sweep of the fused Gaussian ELBO pass (fwd+bwd, VI): algorithmic GB/s = (8*D*B + 4*B) / time."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dun_amd import kernels, _lib as L
from bench import timed
be = kernels.backend()
lib = L.lib()
D = 51
ls = torch.zeros(1, device="cuda"); ql = 0.1 * torch.randn(D, device="cuda")
print("D=%d" % D)
for logB in (13, 16, 18, 20, 22, 23):  # 2^16 = the c3_wide configuration
    B = 1 << logB
    act = torch.randn(D, B, device="cuda"); y = torch.randn(B, device="cuda")
    dact = torch.empty_like(act)
    nparts = lib.dun_ll_num_partials(B, 1)
    partials = torch.empty(D, nparts, device="cuda", dtype=torch.float64)
    s = L.stream_ptr(act.device)
    def run():
        L.check(lib.dun_ll_gauss(L.f32(act), L.f32(y), L.f32(ls), L.f32(ql), 1, D, B, 1.0 / B, L.f32(dact), None,
                                 L.ptr(partials), s))
    ms = timed(run, 20)
    by = 8.0 * D * B + 4.0 * B
    print("B=2^%d  %.1f MB  %.4f ms  %.0f GB/s (%.1f%% of 8 TB/s)" % (logB, by / 1e6, ms, by / ms / 1e6, by / ms / 1e6 / 80), flush=True)

# softmax cross-entropy form (image DUNs): 8*D*B*C + 8*B algorithmic bytes
for C in (10, 100):
    D = 12
    ql = 0.1 * torch.randn(D, device="cuda")
    print("softmax D=%d C=%d" % (D, C))
    for logB in (12, 16, 20, 22 if C == 10 else 20):
        B = 1 << logB
        act = torch.randn(D, B, C, device="cuda"); yy = torch.randint(0, C, (B,), device="cuda")
        dact = torch.empty_like(act)
        nparts = lib.dun_ll_num_partials(B, C)
        partials = torch.empty(D, nparts, device="cuda", dtype=torch.float64)
        s = L.stream_ptr(act.device)
        def run():
            L.check(lib.dun_ll_softmax(L.f32(act), L.ptr(yy), L.f32(ql), 1, D, B, C, 1.0 / B, L.f32(dact), None,
                                       L.ptr(partials), s))
        ms = timed(run, 20)
        by = 8.0 * D * B * C + 8.0 * B
        print("B=2^%d  %.1f MB  %.4f ms  %.0f GB/s (%.1f%% of 8 TB/s)" % (logB, by / 1e6, ms, by / ms / 1e6, by / ms / 1e6 / 80), flush=True)
        del act, dact
