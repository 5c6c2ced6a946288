"""Per-launch times of the first dX GEMM launches after the host-side set-up bench.py does before timing them (GPU only;
profiles/r02_dx_first_launches.txt).  PROBE_MODE=nobn | nofwd selects the variants recorded there."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from bench import WORKLOADS, timed
from dun_amd import kernels
bench.import_torch()
din, W, L, B, N_train, lr, momentum, wd = WORKLOADS["c3_wide"]
net = bench.build_net(din, W, L, N_train, lr, momentum, wd)
x, y = bench.shard_data(B, din, 0, torch.device("cuda", 0))
for _ in range(4):
    net.fit(x, y)
torch.cuda.synchronize()
be = kernels.backend()
model = net.model
buf = model._get_buffers(B, L, True)
buf.ensure_backward()
H0, Z0, w0, b0 = buf.H[0], buf.Z[0], model.stacked["w"][0], model.stacked["b"][0]
scratch = torch.empty_like(Z0)
mode = os.environ.get("PROBE_MODE", "a")
if mode != "nofwd":
    print("fwd %.3f" % timed(lambda: be.gemm(H0, w0, B, W, W, bias=b0, relu=True, colstats=True, out=scratch), 5))
u1 = torch.randn(B, 1, device=Z0.device) * 1e-3
v1 = model.stacked["w_out"]
w0t = w0.t().contiguous()
bn = (Z0, buf.bn_mean[0], buf.bn_rstd[0])
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(17)]
ev[0].record()
for i in range(16):
    be.gemm(buf.dU, w0t, B, W, W, a_kcontig=True, b_kcontig=True, R=buf.G, U=u1, V=v1, out=scratch,
            bn=None if mode == "nobn" else bn)
    ev[i + 1].record()
torch.cuda.synchronize()
print(mode, " ".join("%.2f" % ev[i].elapsed_time(ev[i + 1]) for i in range(16)), flush=True)
