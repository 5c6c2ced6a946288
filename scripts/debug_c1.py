"""Per-layer error growth of the HIP trunk vs an fp64 oracle evaluation (diagnostic, GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import torch.nn.functional as F
from tests.util import Case, rel_err, stacked_grads_from_reference
from tests.test_wrappers_gpu import build_net
from oracle import dun_oracle as O

name = sys.argv[1] if len(sys.argv) > 1 else "c1_wiggle"
c = Case(name); cfg = c.cfg
net = build_net(c)
x, y = c.t("x").cuda(), c.y().cuda()
net.model.train()
act = net.model.forward(x)
buf = list(net.model._fc_cache.values())[0]
o = c.oracle_net(torch.float64); p = O.clone_params(o.p)
x64 = c.t("x").double()
h = F.linear(x64, p["w_in"], p["b_in"])
print("H0", rel_err(buf.H[0].cpu(), h))
for i in range(cfg["L"]):
    u = F.linear(h, p["w"][i], p["b"][i]); z = torch.relu(u)
    m = z.mean(0); v = z.var(0, unbiased=False); rs = 1/torch.sqrt(v+1e-5)
    print(i, "Z %.2e mean %.2e rstd %.2e" % (rel_err(buf.Z[i].cpu(), z), rel_err(buf.bn_mean[i].cpu(), m), rel_err(buf.bn_rstd[i].cpu(), rs)),
          "min var/mean^2 %.2e" % float((v/(m*m+1e-30)).min()), end=" ")
    h = h + (z-m)*rs*p["gamma"][i] + p["beta"][i]
    print("H %.2e" % rel_err(buf.H[i+1].cpu(), h), " ref act err %.2e ours %.2e" % (rel_err(c["act_vec"][i+1], F.linear(h, p["w_out"], p["b_out"])), rel_err(act[i+1].detach().cpu(), F.linear(h, p["w_out"], p["b_out"]))))
prior = net.model.get_w_prior_loglike(k=None)
out = net.prob_model.fused_objective(prior, act, y, net.f_neg_loglike, cfg["N"])
net._backward(act, prior, out)
_, g64, _, _ = o.fit(x64, c.y().double() if cfg["regression"] else c.y(), return_grads=True)
ref = stacked_grads_from_reference(c)
for k in ref:
    if k in ("w", "b", "gamma", "beta"):
        for i in range(cfg["L"]):
            print(k, i, "ours %.2e ref %.2e" % (rel_err(net.model.stacked_grad[k][i].cpu(), g64[k][i]), rel_err(ref[k][i], g64[k][i])))
    else:
        print(k, "ours %.2e ref %.2e" % (rel_err(net.model.stacked_grad[k].cpu(), g64[k]), rel_err(ref[k], g64[k])))
