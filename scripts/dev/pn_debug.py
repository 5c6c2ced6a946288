"""Stage-by-stage comparison of the one-launch small-net kernels against the tiled path (development aid)."""
import sys
import torch
sys.path.insert(0, ".")
from tests.test_smallnet_gpu import make, staged_forward, staged_backward
from dun_amd import _lib

def err(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30)

shape = tuple(int(v) for v in sys.argv[1:7]) if len(sys.argv) >= 7 else (300, 1, 100, 3, 1, 1)
B, din, W, L, Oo, residual = shape
model, x, be, g = make(B, din, W, L, Oo, bool(residual))
model.train()
rm0, rv0 = model.running_mean.clone(), model.running_var.clone()
b1 = be.fc_buffers(model, B, L, True); b1.new_act(); b1.set_input(x)
be.fc_forward(b1)
torch.cuda.synchronize()
print("sync status", _lib.lib().dun_fc_sync_status(b1.c, None), "mode", _lib.lib().dun_fc_small_net_mode(-1))
model.running_mean.copy_(rm0); model.running_var.copy_(rv0)
b2 = be.fc_buffers(model, B, L, True); b2.new_act(); b2.set_input(x)
staged_forward(be, b2, L, B)
for d in range(L + 1):
    print("H[%d]" % d, err(b1.H[d], b2.H[d]), "act", err(b1.act[d], b2.act[d]))
    if d < L:
        print("  Z[%d]" % d, err(b1.Z[d], b2.Z[d]), "mean", err(b1.bn_mean[d], b2.bn_mean[d]), "rstd", err(b1.bn_rstd[d], b2.bn_rstd[d]))
        bad = (b1.Z[d] - b2.Z[d]).abs() > 1e-3
        if bad.any():
            idx = bad.nonzero()
            print("   bad Z entries:", idx.shape[0], "rows", idx[:, 0].unique()[:20].tolist(), "cols", idx[:, 1].unique()[:40].tolist())
for k in ("H", "Z", "bn_mean", "bn_rstd"):
    getattr(b2, k).copy_(getattr(b1, k))
dact = (torch.randn(L + 1, B, Oo, generator=g) / B).cuda()
b1.ensure_backward(); b1.set_dact(dact)
model.flat_grads.fill_(float("nan"))
be.fc_backward(b1)
g1 = {k: v.clone() for k, v in model.stacked_grad.items()}
G1 = b1.G.clone()
b2.ensure_backward(); b2.set_dact(dact)
model.flat_grads.fill_(float("nan"))
staged_backward(be, b2, L, B)
for k, v in model.stacked_grad.items():
    print("grad", k, err(g1[k], v), bool(torch.isfinite(g1[k]).all()))
print("G", err(G1, b2.G))
print("sync status", _lib.lib().dun_fc_sync_status(b1.c, None))
z = b2.Z[0].double()
var = z.var(0, unbiased=False)
print("ours var", (1.0 / b1.bn_rstd[0].double() ** 2 - 1e-5)[:6].tolist())
print("true var", var[:6].tolist())
print("E[z^2]  ", (z * z).mean(0)[:6].tolist())
mp = torch.stack([z[16 * p:16 * p + 16].mean(0) for p in range((B + 15) // 16)])
npw = torch.tensor([min(16, B - 16 * p) for p in range((B + 15) // 16)], dtype=torch.float64, device=z.device)
print("between ", ((npw[:, None] * mp * mp).sum(0) / B - z.mean(0) ** 2)[:6].tolist())
