"""Shared helpers for the parity tests: golden-fixture loading and oracle construction."""
import json
import os

import numpy as np
import torch

from oracle import dun_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

FC_CASES = ["c1_wiggle", "c2_boston", "c2_protein", "c3_reduced", "mll_small", "mlp_nores", "fixed_sig",
            "cls_fc", "cls_fc_mll"]
# weight-prior path: the reference's diag_w_Gauss_loglike, plain / BMA_prior, VI / exact marginal
PRIOR_CASES = ["prior_gauss", "prior_bma", "prior_mll", "prior_cls"]


class Case(object):
    def __init__(self, name):
        self.name = name
        self.z = np.load(os.path.join(GOLDEN, name + ".npz"))
        self.cfg = json.loads(str(self.z["meta"])) if "meta" in self.z.files else {}

    def __getitem__(self, k):
        return self.z[k]

    def has(self, k):
        return k in self.z.files

    def t(self, k):
        return torch.from_numpy(np.array(self.z[k]))

    def sub(self, prefix):
        return {k[len(prefix):]: self.z[k] for k in self.z.files if k.startswith(prefix)}

    @property
    def D(self):
        return self.cfg["L"] + 1

    def prior(self):
        return [1.0 / self.D] * self.D

    def y(self):
        y = self.t("y")
        return y if self.cfg["regression"] else y.long()

    def w_prior_fn(self):
        """The case's weight prior as the oracle's callable (None for the w_prior=None cases)."""
        wp = self.cfg.get("w_prior")
        return None if not wp else O.diag_gauss_prior(wp["mu"], wp["sig2"], wp["mult"])

    def oracle_net(self, dtype=torch.float32):
        cfg = self.cfg
        p = O.params_from_reference_state_dict(self.sub("sd/"), cfg["L"], dtype=dtype)
        net = O.OracleNet(p, self.prior(), cfg["N"], lr=cfg["lr"], momentum=cfg["momentum"],
                          weight_decay=cfg["wd"], regression=cfg["regression"], vi=cfg["vi"],
                          residual=not cfg.get("mlp", False), pred_sig=cfg.get("pred_sig"),
                          w_prior=self.w_prior_fn(), bma=bool(cfg.get("bma")))
        if cfg["vi"]:
            net.q_logits = self.t("q_logits0").to(dtype)
        if cfg["regression"]:
            net.log_std = self.t("log_std0").to(dtype)
        return net


class TestGaussPrior(object):
    """A weight prior as a caller would hand it to the model constructors: any callable tensor -> scalar.  Defined
    here, on purpose, instead of importing the product's diag_w_Gauss_loglike: the wrappers must work with any such
    callable.  Same density as the reference's diag_w_Gauss_loglike (its sigma^2 argument is the Normal's scale)."""
    __test__ = False

    def __init__(self, mu, scale, mult):
        self.mu, self.scale, self.mult = mu, scale, mult

    def __call__(self, w):
        import math
        lp = -((w - self.mu) ** 2) / (2 * self.scale ** 2) - math.log(self.scale) - 0.5 * math.log(2 * math.pi)
        return lp.sum() * self.mult


def model_prior_kwargs(case):
    """Constructor kwargs (w_prior, BMA_prior) of a golden case."""
    wp = case.cfg.get("w_prior")
    return dict(w_prior=None if not wp else TestGaussPrior(wp["mu"], wp["sig2"], wp["mult"]),
                BMA_prior=bool(case.cfg.get("bma")))


def stacked_grads_from_reference(case, prefix="g/"):
    """Reference per-module grads -> the stacked layout of oracle/dun_amd params."""
    g = case.sub(prefix)
    L = case.cfg["L"]
    out = {"w_in": g["input_layer.weight"], "b_in": g["input_layer.bias"],
           "w_out": g["output_layer.weight"], "b_out": g["output_layer.bias"]}
    pre = "stochstic_layers.%d.block.%d.%s"
    if L:
        out["w"] = np.stack([g[pre % (i, 0, "weight")] for i in range(L)])
        out["b"] = np.stack([g[pre % (i, 0, "bias")] for i in range(L)])
        out["gamma"] = np.stack([g[pre % (i, 2, "weight")] for i in range(L)])
        out["beta"] = np.stack([g[pre % (i, 2, "bias")] for i in range(L)])
    return out


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    denom = max(np.abs(b).max(), 1e-30)
    return float(np.abs(a - b).max() / denom)


def oracle_grads_with_masks(case, masks, dtype=torch.float64):
    """Gradients of the case's loss from an fp64 evaluation of the oracle in which every ReLU uses the given
    activation pattern (masks [L,B,W], bool).  fp32 implementations legitimately disagree on the sign of
    pre-activations within rounding noise of zero; one such flip moves a weight gradient by O(1/B), far above
    fp32 rounding.  Forcing the pattern isolates the arithmetic under test from that coin toss."""
    import torch.nn.functional as F
    cfg = case.cfg
    net = case.oracle_net(dtype)
    p = net.p
    leaves = {k: p[k] for k in O.PARAM_KEYS}
    if cfg["vi"]:
        leaves["q_logits"] = net.q_logits
    if cfg["regression"] and cfg.get("pred_sig") is None:
        leaves["log_std"] = net.log_std
    for v in leaves.values():
        v.requires_grad_(True)
    x = case.t("x").to(dtype)
    y = case.y() if not cfg["regression"] else case.y().to(dtype)
    residual = not cfg.get("mlp", False)
    h = F.linear(x, p["w_in"], p["b_in"])
    acts = [F.linear(h, p["w_out"], p["b_out"])]
    for i in range(cfg["L"]):
        u = F.linear(h, p["w"][i], p["b"][i])
        z = u * masks[i].to(dtype)
        m, v = z.mean(0), z.var(0, unbiased=False)
        blk = (z - m) / torch.sqrt(v + 1e-5) * p["gamma"][i] + p["beta"][i]
        h = h + blk if residual else blk
        acts.append(F.linear(h, p["w_out"], p["b_out"]))
    act_vec = torch.stack(acts)
    D = cfg["L"] + 1
    prior = net.prior_loglikes()  # differentiable w.r.t. the leaves: net.p holds them
    joint = O.joint_loglike(prior, act_vec, y, cfg["regression"], cfg["N"], net.log_std)
    if cfg["vi"]:
        obj = (O.q_probs(net.q_logits) * joint).sum() - O.kl_q_prior(net.q_logits, net.logprior)
    else:
        obj = O.marg_loglike(joint, net.logprior)
    grads = torch.autograd.grad(-obj / cfg["N"], list(leaves.values()))
    return dict(zip(leaves.keys(), grads)), act_vec.detach()
