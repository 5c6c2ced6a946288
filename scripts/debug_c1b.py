"""Staged backward: dU_i and G_i of the HIP trunk vs fp64 autograd (diagnostic, GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import torch.nn.functional as F
from tests.util import Case, rel_err
from tests.test_wrappers_gpu import build_net
from oracle import dun_oracle as O
from dun_amd import kernels

name = sys.argv[1] if len(sys.argv) > 1 else "c1_wiggle"
c = Case(name); cfg = c.cfg; L = cfg["L"]
net = build_net(c)
x, y = c.t("x").cuda(), c.y().cuda()
net.model.train()
act = net.model.forward(x)
buf = list(net.model._fc_cache.values())[0]
prior = net.model.get_w_prior_loglike(k=None)
out = net.prob_model.fused_objective(prior, act, y, net.f_neg_loglike, cfg["N"])
# fp64 truth with retained intermediates
o = c.oracle_net(torch.float64); p = O.clone_params(o.p)
for k in O.PARAM_KEYS: p[k].requires_grad_(True)
x64 = c.t("x").double(); y64 = c.y().double()
h = F.linear(x64, p["w_in"], p["b_in"]); hs=[h]; us=[]; acts=[F.linear(h,p["w_out"],p["b_out"])]
for i in range(L):
    u = F.linear(h, p["w"][i], p["b"][i]); u.retain_grad(); us.append(u)
    z = torch.relu(u); m = z.mean(0); v = z.var(0, unbiased=False)
    h = h + (z-m)/torch.sqrt(v+1e-5)*p["gamma"][i] + p["beta"][i]; h.retain_grad(); hs.append(h)
    acts.append(F.linear(h,p["w_out"],p["b_out"]))
hs[0].retain_grad()
av = torch.stack(acts)
e = O.elbo(torch.zeros(L+1).double(), av, y64, True, cfg["N"], o.q_logits, o.logprior, o.log_std)
(-e/cfg["N"]).backward()
be = kernels.backend()
buf.ensure_backward(); buf.set_dact(out.dact.contiguous())
be.fc_bwd_begin(buf)
print("G_L", rel_err(buf.G.cpu(), hs[L].grad))
for i in range(L-1, -1, -1):
    be.fc_bwd_block_reduce(buf, i)
    be.fc_bwd_block_apply(buf, i, buf.B)
    dU = buf.dU.cpu().double(); t = us[i].grad
    diff = (dU - t).abs()
    mism = ((dU != 0) != (t != 0)).sum().item()
    print(i, "dU err %.2e  mask mismatches %d  worst (b,n) %s  G err %.2e" % (rel_err(dU, t), mism, np.unravel_index(int(diff.argmax()), diff.shape), rel_err(buf.G.cpu(), hs[i].grad)),
          " u at mismatches:", us[i].detach()[((dU != 0) != (t != 0))][:5].numpy())
