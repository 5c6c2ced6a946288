"""Trunk forward / backward time of a small DUN (default: the UCI shape) through dun_fc_forward / dun_fc_backward.
Run once as is (column-owned small-net kernels) and once with DUN_SMALLNET=0 (tiled-GEMM path) to compare."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dun_amd import kernels  # noqa: E402
from dun_amd.stochastic_fc_models import arq_uncert_fc_resnet  # noqa: E402


def timed(fn, n=200):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    B, din, W, L = [int(a) for a in sys.argv[1:5]] if len(sys.argv) >= 5 else (512, 13, 100, 10)
    torch.manual_seed(0)
    model = arq_uncert_fc_resnet(din, 1, W, L).cuda()
    model.train()
    be = kernels.backend()
    buf = be.fc_buffers(model, B, L, True)
    buf.new_act()
    buf.set_input(torch.randn(B, din, device="cuda"))
    buf.ensure_backward()
    buf.set_dact(torch.randn(L + 1, B, 1, device="cuda") / B)
    fwd = timed(lambda: be.fc_forward(buf))
    bwd = timed(lambda: be.fc_backward(buf))
    # the same launch sequences replayed from a hipGraph: kernel time without the host's launch cost
    out = {}
    for name, fn in (("fwd", lambda: be.fc_forward(buf)), ("bwd", lambda: be.fc_backward(buf))):
        g = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            fn()
        out[name] = timed(g.replay)
    print("smallnet=%s B=%d din=%d W=%d L=%d: forward %.1f us (graph %.1f), backward %.1f us (graph %.1f)" %
          (os.environ.get("DUN_SMALLNET", "1"), B, din, W, L, fwd, out["fwd"], bwd, out["bwd"]), flush=True)


if __name__ == "__main__":
    main()
