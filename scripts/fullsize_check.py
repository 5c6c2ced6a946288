"""Exploration: one full-size (B=65536, W=2048) train-mode forward + objective + backward of dun_amd against the
oracle's functions evaluated ON THE GPU (torch's own fp32 / fp64 kernels).  Prints the relative differences."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from oracle import dun_oracle as O

L = int(sys.argv[1]) if len(sys.argv) > 1 else 4
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
dt64 = len(sys.argv) > 3 and sys.argv[3] == "f64"
din, W, N = 64, 2048, 1 << 20
net = bench.build_net(din, W, L, N, 1e-4, 0.5, 0.0)
x, y = bench.shard_data(B, din, 0, torch.device("cuda"))
net.model.train()
act = net.model.forward(x)
prior = net.model.get_w_prior_loglike(k=None)
out = net.prob_model.fused_objective(prior, act, y, net.f_neg_loglike, N)
net._backward(act, prior, out)
torch.cuda.synchronize()


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def rel2(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


for dt in ([torch.float32, torch.float64] if dt64 else [torch.float32]):
    p = {k: net.model.stacked[k].detach().to(dt).clone().requires_grad_(True) for k in O.PARAM_KEYS}
    p["running_mean"] = torch.zeros(L, W, device="cuda", dtype=dt)
    p["running_var"] = torch.ones(L, W, device="cuda", dtype=dt)
    p["num_batches_tracked"] = torch.zeros(L, dtype=torch.int64, device="cuda")
    ql = net.prob_model.q_logits.detach().to(dt).clone().requires_grad_(True)
    ls = net.f_neg_loglike.log_std.detach().to(dt).clone().requires_grad_(True)
    D = L + 1
    logprior = torch.full((D,), 1.0 / D, device="cuda", dtype=dt).log()
    a = O.fc_forward(p, x.to(dt), train=True)
    joint = O.joint_loglike(torch.zeros(D, device="cuda", dtype=dt), a, y.to(dt), True, N, ls)
    elbo = (O.q_probs(ql) * joint).sum() - O.kl_q_prior(ql, logprior)
    leaves = [p[k] for k in O.PARAM_KEYS] + [ql, ls]
    g = torch.autograd.grad(-elbo / N, leaves)
    g = dict(zip(list(O.PARAM_KEYS) + ["q_logits", "log_std"], g))
    print("== oracle on GPU, %s, L=%d B=%d" % (dt, L, B))
    print("act_vec  max-rel %.3e" % rel(act.detach(), a.detach()))
    print("joint    max-rel %.3e" % rel(out.joint, joint.detach()))
    print("ELBO     rel     %.3e   (ours %.6e, oracle %.6e)" % (rel(out.objective, elbo.detach()), float(out.objective), float(elbo)))
    for k in O.PARAM_KEYS:
        print("grad %-6s max-rel %.3e  l2-rel %.3e" % (k, rel(net.model.stacked_grad[k], g[k]), rel2(net.model.stacked_grad[k], g[k])))
    print("grad q_logits max-rel %.3e" % rel(net.prob_model.q_logits.grad, g["q_logits"]))
    print("grad log_std  max-rel %.3e" % rel(net.f_neg_loglike.log_std.grad, g["log_std"]), flush=True)
    del p, a, joint, elbo, g, leaves
    torch.cuda.empty_cache()
