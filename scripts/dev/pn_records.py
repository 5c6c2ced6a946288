import sys
import torch
sys.path.insert(0, ".")
from tests.test_smallnet_gpu import make
B, din, W, L = 300, 1, 100, 1
model, x, be, g = make(B, din, W, L, 1, True)
model.train()
b1 = be.fc_buffers(model, B, L, True); b1.new_act(); b1.set_input(x)
be.fc_forward(b1)
torch.cuda.synchronize()
P = (B + 15) // 16
words = b1.sync.view(torch.int32)[64:64 + P * 128 * 4].view(P, 128, 4)
a = words[:, :, 0].contiguous().view(torch.float32)
bb = words[:, :, 2].contiguous().view(torch.float32)
print("header", b1.sync[:4].tolist(), "tags", words[:2, :3, 1].tolist(), words[:2, :3, 3].tolist())
z = b1.Z[0].double()
for p in (0, 1, 18):
    zz = z[16 * p:16 * p + 16]
    mp = zz.mean(0)
    m2 = ((zz - mp) ** 2).sum(0)
    print("p", p, "mean ours", a[p, :4].tolist(), "true", mp[:4].tolist())
    print("      M2 ours", bb[p, :4].tolist(), "true", m2[:4].tolist())
npw = torch.tensor([min(16, B - 16 * p) for p in range(P)], dtype=torch.float64, device=z.device)
A, Bm = a.double(), bb.double()
t1 = (npw[:, None] * A).sum(0); t2 = (Bm + npw[:, None] * A * A).sum(0)
mean = t1 / B; var = t2 / B - mean * mean
print("var from records", var[:6].tolist())
print("ours var        ", (1.0 / b1.bn_rstd[0].double() ** 2 - 1e-5)[:6].tolist())
print("true var        ", z.var(0, unbiased=False)[:6].tolist())
print("ours mean", b1.bn_mean[0][:6].tolist(), "true", z.mean(0)[:6].tolist())
# hypotheses
print("no-n  ", ((Bm + A * A).sum(0) / B - mean * mean)[:6].tolist())
print("only b/n", (Bm.sum(0) / B)[:6].tolist())
