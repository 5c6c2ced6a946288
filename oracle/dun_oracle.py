"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the DUN depth-marginalised train/predict path.

A plain-torch (CPU) restatement of what the reference computes on this path, written functionally over a
dict of stacked parameter tensors.  It is the checker for the HIP kernels and the ``cpu_baseline`` leg of
``bench.py``; the product package never imports it.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the real reference from
``/root/reference`` in the build container, runs it on seeded inputs and commits the inputs/outputs as
``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks every function below against those vectors.
(The reference itself ships no tests or golden vectors -- SURVEY.md F6.)
The functions are device-agnostic plain torch: the full-size parity tests (tests/test_fullsize_gpu.py) evaluate them on
device tensors (torch's own fp32/fp64 kernels) where the CPU would need minutes; it remains the checker there.

Reference locations each function follows (paths relative to the reference repo):
  fc_forward ............ src/DUN/stochastic_fc_models.py:62-70, src/DUN/layers.py:33-53
  fc_vec_forward ........ src/DUN/stochastic_fc_models.py:54-60
  gauss_nll ............. src/probability.py:54-60
  ce_rows ............... torch CrossEntropyLoss(reduction='none'), src/DUN/training_wrappers.py:29
  joint_loglike ......... src/probability.py:162-174
  marg_loglike .......... src/probability.py:176-179
  depth_log_posterior ... src/probability.py:181-186
  q_probs/q_logprobs/kl . src/probability.py:224-238
  elbo .................. src/probability.py:240-253
  marginalise_predict ... src/probability.py:188-208
  sgd_step .............. torch.optim.SGD as configured at src/DUN/training_wrappers.py:48-52
  fit_vi / fit_mll ...... src/DUN/training_wrappers.py:212-245 / :54-89
  evaluate / predict .... src/DUN/training_wrappers.py:91-115 / :136-158
  elbo_grads_closed_form  hand-derived backward, SURVEY.md section 8 row a11
  diag_gauss_prior ...... src/probability.py:30-40 (diag_w_Gauss_loglike)
  w_prior_loglike ....... src/DUN/stochastic_fc_models.py:26-52 (apply_prior / get_w_prior_loglike)
"""
import math

import torch
import torch.nn.functional as F

LOG_SQRT_2PI = 0.5 * math.log(2.0 * math.pi)


# --------------------------------------------------------------------------------------------------
# parameters
# --------------------------------------------------------------------------------------------------
PARAM_KEYS = ("w_in", "b_in", "w_out", "b_out", "w", "b", "gamma", "beta")
BUFFER_KEYS = ("running_mean", "running_var", "num_batches_tracked")


def init_fc_params(input_dim, output_dim, width, n_layers, generator=None, dtype=torch.float32):
    """Default torch.nn init (kaiming-uniform(a=sqrt 5) Linear, BN gamma=1 beta=0), stacked layout."""
    def lin(out_f, in_f):
        bound = 1.0 / math.sqrt(in_f)
        w = (torch.rand(out_f, in_f, generator=generator, dtype=dtype) * 2 - 1) * bound
        b = (torch.rand(out_f, generator=generator, dtype=dtype) * 2 - 1) * bound
        return w, b
    p = {}
    p["w_in"], p["b_in"] = lin(width, input_dim)
    p["w_out"], p["b_out"] = lin(output_dim, width)
    ws, bs = [], []
    for _ in range(n_layers):
        w, b = lin(width, width)
        ws.append(w)
        bs.append(b)
    p["w"] = torch.stack(ws) if n_layers else torch.zeros(0, width, width, dtype=dtype)
    p["b"] = torch.stack(bs) if n_layers else torch.zeros(0, width, dtype=dtype)
    p["gamma"] = torch.ones(n_layers, width, dtype=dtype)
    p["beta"] = torch.zeros(n_layers, width, dtype=dtype)
    p["running_mean"] = torch.zeros(n_layers, width, dtype=dtype)
    p["running_var"] = torch.ones(n_layers, width, dtype=dtype)
    p["num_batches_tracked"] = torch.zeros(n_layers, dtype=torch.int64)
    return p


def params_from_reference_state_dict(sd, n_layers, dtype=torch.float32):
    """Stack a reference ``arq_uncert_fc_resnet.state_dict()`` (torch tensors or numpy arrays)."""
    def t(k):
        v = sd[k]
        return torch.as_tensor(v).clone()
    p = {"w_in": t("input_layer.weight").to(dtype), "b_in": t("input_layer.bias").to(dtype),
         "w_out": t("output_layer.weight").to(dtype), "b_out": t("output_layer.bias").to(dtype)}
    W = p["w_in"].shape[0]
    pre = "stochstic_layers.%d.block.%d.%s"
    def stack(fmt_idx, name, shape):
        if n_layers == 0:
            return torch.zeros(*shape, dtype=dtype)
        return torch.stack([t(pre % (i, fmt_idx, name)).to(dtype) for i in range(n_layers)])
    p["w"] = stack(0, "weight", (0, W, W))
    p["b"] = stack(0, "bias", (0, W))
    p["gamma"] = stack(2, "weight", (0, W))
    p["beta"] = stack(2, "bias", (0, W))
    p["running_mean"] = stack(2, "running_mean", (0, W))
    p["running_var"] = stack(2, "running_var", (0, W))
    if n_layers:
        p["num_batches_tracked"] = torch.stack(
            [t(pre % (i, 2, "num_batches_tracked")).to(torch.int64) for i in range(n_layers)])
    else:
        p["num_batches_tracked"] = torch.zeros(0, dtype=torch.int64)
    return p


def clone_params(p):
    return {k: v.clone() for k, v in p.items()}


# --------------------------------------------------------------------------------------------------
# trunk: residual MLP emitting every depth through the one shared head
# --------------------------------------------------------------------------------------------------
def _block(p, i, h, train, momentum, eps):
    u = F.linear(h, p["w"][i], p["b"][i])
    z = torch.relu(u)
    if train:
        # F.batch_norm updates the running buffers in place when handed them (momentum 0.1, unbiased var)
        rm, rv = p["running_mean"][i], p["running_var"][i]
        out = F.batch_norm(z, rm, rv, p["gamma"][i], p["beta"][i], True, momentum, eps)
        p["num_batches_tracked"][i] += 1
        return out
    return F.batch_norm(z, p["running_mean"][i], p["running_var"][i], p["gamma"][i], p["beta"][i],
                        False, momentum, eps)


def fc_forward(p, x, depth=None, train=True, residual=True, momentum=0.1, eps=1e-5, return_hidden=False):
    """act_vec[D,B,O]: head(h_0), head(h_1), ... with h_0 = input_layer(x) (no nonlinearity),
    h_{i+1} = h_i + BN(ReLU(Linear_i(h_i))) (residual) or BN(ReLU(Linear_i(h_i))) (plain MLP)."""
    n_layers = p["w"].shape[0]
    depth = n_layers if depth is None else depth
    h = F.linear(x, p["w_in"], p["b_in"])
    acts = [F.linear(h, p["w_out"], p["b_out"])]
    hidden = [h]
    for i in range(depth):
        blk = _block(p, i, h, train, momentum, eps)
        h = h + blk if residual else blk
        hidden.append(h)
        acts.append(F.linear(h, p["w_out"], p["b_out"]))
    act_vec = torch.stack(acts, dim=0)
    if return_hidden:
        return act_vec, torch.stack(hidden, dim=0)
    return act_vec


def fc_vec_forward(p, x, vec, residual=True, eps=1e-5):
    """Final-head output with an explicit 0/1 mask per block (eval-mode BN, as the only caller uses it)."""
    n_layers = p["w"].shape[0]
    assert vec.shape[0] == n_layers
    h = F.linear(x, p["w_in"], p["b_in"])
    for i in range(n_layers):
        blk = _block(p, i, h, False, 0.1, eps)
        h = h + vec[i] * blk if residual else blk
    return F.linear(h, p["w_out"], p["b_out"])


# --------------------------------------------------------------------------------------------------
# likelihoods
# --------------------------------------------------------------------------------------------------
def gauss_nll(mu, y, log_std, model_std=None):
    sig = log_std.exp().clamp(min=1e-4)
    if model_std is not None:
        sig = (sig ** 2 + model_std ** 2) ** 0.5
    return LOG_SQRT_2PI + torch.log(sig) + (y - mu) ** 2 / (2 * sig ** 2)


def ce_rows(logits, y, ignore_index=-100):
    """CrossEntropyLoss(reduction='none'): rows labelled ``ignore_index`` give 0 (and no gradient); any other label
    outside [0, C) raises, as torch does on CPU."""
    C = logits.shape[1]
    keep = y != ignore_index
    if bool(((y < 0) | (y >= C))[keep].any()):
        raise IndexError("Target out of bounds")
    safe = torch.where(keep, y, torch.zeros_like(y))
    rows = torch.logsumexp(logits, dim=1) - logits.gather(1, safe.view(-1, 1)).squeeze(1)
    return torch.where(keep, rows, torch.zeros_like(rows))


def loglike_per_act(act_vec, y, regression, log_std=None):
    """loglike[d,b] (the quantity the north-star's 1e-5 parity bar is stated on)."""
    D, B = act_vec.shape[0], act_vec.shape[1]
    if regression:
        flat = act_vec.reshape(D * B, -1)
        y_rep = y.repeat(D, 1)
        # like the reference this only has a [D,B] shape for one output dimension
        return (-gauss_nll(flat, y_rep, log_std)).view(D, B)
    flat = act_vec.reshape(D * B, -1)
    return (-ce_rows(flat, y.repeat(D))).view(D, B)


def joint_loglike(prior_loglikes, act_vec, y, regression, N_train, log_std=None):
    B = act_vec.shape[1]
    ll = loglike_per_act(act_vec, y, regression, log_std)
    return (N_train / B) * ll.sum(dim=1) + prior_loglikes


# --------------------------------------------------------------------------------------------------
# weight priors
# --------------------------------------------------------------------------------------------------
def diag_gauss_prior(mu=0.0, sig2=1.0, multiplier=1.0):
    """log N(w; mu, scale) summed over a tensor.  NOTE: the reference hands its ``sig2`` argument to
    torch.distributions.Normal as the SCALE (standard deviation), not the variance; restated as it behaves."""
    def f(w):
        lp = -((w - mu) ** 2) / (2 * sig2 ** 2) - math.log(sig2) - math.log(math.sqrt(2 * math.pi))
        return lp.sum() * multiplier
    return f


def w_prior_loglike(p, prior_f, bma=False, k=None):
    """Scalar the reference's get_w_prior_loglike returns when a weight prior is set: per-layer sums of the prior
    over Linear weights and biases (input + shared output layer form entry 0, block i entry i+1; BatchNorm
    parameters carry no prior), optionally turned into cumulative sums (BMA_prior), then summed over depth."""
    n_layers = p["w"].shape[0]
    k = n_layers if k is None else k
    first = prior_f(p["w_in"]) + prior_f(p["b_in"]) + prior_f(p["w_out"]) + prior_f(p["b_out"])
    per_layer = [first] + [prior_f(p["w"][i]) + prior_f(p["b"][i]) for i in range(k)]
    vec = torch.stack(per_layer, dim=0)
    if bma:
        vec = vec.unsqueeze(0).repeat(k + 1, 1).tril(diagonal=0).sum(dim=1)
    return vec.sum(dim=0)


# --------------------------------------------------------------------------------------------------
# depth categorical: exact marginal and variational posterior
# --------------------------------------------------------------------------------------------------
def marg_loglike(joint, logprior):
    return torch.logsumexp(joint + logprior, dim=0)


def depth_log_posterior(joint, logprior, log_marg=None):
    if log_marg is None:
        log_marg = marg_loglike(joint, logprior)
    return joint + logprior - log_marg


def q_logprobs(q_logits):
    return F.log_softmax(q_logits, dim=0)


def q_probs(q_logits):
    return F.softmax(q_logits, dim=0)


def kl_q_prior(q_logits, logprior, eps=1e-35):
    log_q = q_logprobs(q_logits)
    q = q_probs(q_logits).clamp(min=eps, max=(1 - eps))
    return (q * (log_q - logprior)).sum(dim=0)


def elbo(prior_loglikes, act_vec, y, regression, N_train, q_logits, logprior, log_std=None, beta=1.0,
         eps=1e-35):
    joint = joint_loglike(prior_loglikes, act_vec, y, regression, N_train, log_std)
    e_ll = (q_probs(q_logits) * joint).sum(dim=0)
    return e_ll - beta * kl_q_prior(q_logits, logprior, eps)


def marginalise_predict(act_vec, d_posterior, softmax=False, get_std=False):
    assert not (softmax and get_std)
    preds = F.softmax(act_vec, dim=2) if softmax else act_vec
    q = d_posterior.clone().detach()
    while q.dim() < act_vec.dim():
        q = q.unsqueeze(1)
    if get_std:
        mu = (q * preds).sum(dim=0)
        var = (q * preds ** 2).sum(dim=0) - mu ** 2
        return mu, var.pow(0.5)
    return (q * preds).sum(dim=0)


def elbo_grads_closed_form(act_vec, y, regression, N_train, q_or_post, log_std=None, vi=True,
                           q_logits=None, logprior=None, prior_loglikes=None, beta=1.0, eps=1e-35):
    """Closed-form d(loss)/d(act, q_logits, log_std) for loss = -ELBO/N (vi) or -log_marg/N (mll).

    dact[d,b] = -(w_d/B) * dll/dact, w = q (VI) or the exact posterior (MLL);
    dll/dmu = (y-mu)/sig^2 (regression), onehot(y)-softmax(a) (classification).
    dlog_std = -(1/B) sum_d w_d sum_b [(y-mu)^2/sig^2 - 1]  (0 under the 1e-4 clamp).
    dq_logits[k] = -(1/N) q_k [(joint_k - E) - beta((log q_k - log p_k) - KL)]  (q inside the eps clamp).
    """
    D, B = act_vec.shape[0], act_vec.shape[1]
    w = q_or_post.view(D, *([1] * (act_vec.dim() - 1)))
    out = {}
    if regression:
        sig_raw = log_std.exp()
        sig = sig_raw.clamp(min=1e-4)
        r = y.unsqueeze(0) - act_vec
        out["dact"] = -(w / B) * r / sig ** 2
        live = (sig_raw >= 1e-4).to(act_vec.dtype)
        out["dlog_std"] = -(1.0 / B) * (w * (r ** 2 / sig ** 2 - 1.0)).sum(dim=(0, 1)) * live
    else:
        sm = F.softmax(act_vec, dim=2)
        keep = (y != -100)
        oh = F.one_hot(torch.where(keep, y, torch.zeros_like(y)), act_vec.shape[2]).to(act_vec.dtype).unsqueeze(0)
        out["dact"] = -(w / B) * (oh - sm) * keep.view(1, B, 1).to(act_vec.dtype)  # ignored rows: no gradient
    if vi and q_logits is not None:
        joint = joint_loglike(prior_loglikes, act_vec, y, regression, N_train, log_std)
        q = q_probs(q_logits)
        lq = q_logprobs(q_logits)
        E = (q * joint).sum()
        KL = (q * (lq - logprior)).sum()
        out["dq_logits"] = -(1.0 / N_train) * q * ((joint - E) - beta * ((lq - logprior) - KL))
    return out


# --------------------------------------------------------------------------------------------------
# optimiser and full steps
# --------------------------------------------------------------------------------------------------
def sgd_step(params, grads, bufs, lr, momentum, weight_decay):
    """torch.optim.SGD (dampening 0, no nesterov): g += wd*p; buf = m*buf + g (buf=g first); p -= lr*buf."""
    for k, p in params.items():
        g = grads.get(k)
        if g is None:
            continue
        if weight_decay != 0:
            g = g + weight_decay * p
        if momentum != 0:
            if k not in bufs:
                bufs[k] = g.clone()
            else:
                bufs[k].mul_(momentum).add_(g)
            g = bufs[k]
        p.sub_(lr * g)


def _train_metrics(act_vec, y, posterior, regression, log_std):
    if regression:
        mu, mstd = marginalise_predict(act_vec, posterior, softmax=False, get_std=True)
        nll = gauss_nll(mu, y, log_std, model_std=mstd).mean(dim=0)
        err = F.mse_loss(mu, y, reduction="mean").sqrt().item()
        return nll.item(), err
    probs = marginalise_predict(act_vec, posterior, softmax=True)
    nll = F.nll_loss(torch.log(probs), y, reduction="none").mean(dim=0)
    err = probs.max(dim=1)[1].ne(y).sum().item() / y.shape[0]
    return nll.item(), err


class OracleNet(object):
    """State + fit/eval/predict of a DUN over the FC trunk; mirrors DUN / DUN_VI numerics on CPU."""

    def __init__(self, params, prior_probs, N_train, lr=1e-2, momentum=0.5, weight_decay=0.0,
                 regression=True, vi=True, residual=True, pred_sig=None, train_q=True, w_prior=None, bma=False):
        self.p = params
        self.w_prior, self.bma = w_prior, bma  # callable on a tensor (e.g. diag_gauss_prior(...)) or None
        dt, dev = params["w_in"].dtype, params["w_in"].device  # (state lives where the parameters live)
        self.prior_probs = torch.as_tensor(prior_probs, dtype=dt, device=dev)
        self.logprior = self.prior_probs.log()
        self.D = self.prior_probs.shape[0]
        self.q_logits = torch.zeros(self.D, dtype=dt, device=dev)
        O = params["w_out"].shape[0]
        self.log_std = (torch.zeros(O, dtype=dt, device=dev) if pred_sig is None
                        else torch.ones(O, dtype=dt, device=dev) * math.log(pred_sig))
        self.train_log_std = regression and pred_sig is None
        self.train_q = train_q and vi
        self.N = N_train
        self.lr, self.momentum, self.wd = lr, momentum, weight_decay
        self.regression, self.vi, self.residual = regression, vi, residual
        self.bufs = {}
        self.current_posterior = None

    def _leaves(self):
        leaves = {k: self.p[k] for k in PARAM_KEYS}
        if self.train_q:
            leaves["q_logits"] = self.q_logits
        if self.train_log_std:
            leaves["log_std"] = self.log_std
        return leaves

    def objective(self, x, y, train=True):
        """(objective, joint[D], act_vec) with autograd enabled on whatever leaves require grad."""
        act_vec = fc_forward(self.p, x, train=train, residual=self.residual)
        prior = self.prior_loglikes()
        joint = joint_loglike(prior, act_vec, y, self.regression, self.N, self.log_std)
        if self.vi:
            e_ll = (q_probs(self.q_logits) * joint).sum()
            obj = e_ll - kl_q_prior(self.q_logits, self.logprior)
        else:
            obj = marg_loglike(joint, self.logprior)
        return obj, joint, act_vec

    def prior_loglikes(self):
        if self.w_prior is None:
            return torch.zeros(self.D, dtype=self.p["w_in"].dtype, device=self.p["w_in"].device)
        return w_prior_loglike(self.p, self.w_prior, self.bma)

    def fit(self, x, y, return_grads=False):
        leaves = self._leaves()
        for v in leaves.values():
            v.requires_grad_(True)
        obj, joint, act_vec = self.objective(x, y, train=True)
        loss = -obj / self.N
        grads_list = torch.autograd.grad(loss, list(leaves.values()), allow_unused=True)
        for v in leaves.values():
            v.requires_grad_(False)
        grads = dict(zip(leaves.keys(), grads_list))
        with torch.no_grad():
            sgd_step(leaves, grads, self.bufs, self.lr, self.momentum, self.wd)
            act_vec = act_vec.detach()
            joint = joint.detach()
            if self.vi:
                self.current_posterior = q_probs(self.q_logits)
            else:
                self.current_posterior = depth_log_posterior(joint, self.logprior).exp()
            nll, err = _train_metrics(act_vec, y, self.current_posterior, self.regression, self.log_std)
        out = (obj.item(), nll, err)
        if return_grads:
            return out, grads, act_vec, joint
        return out

    def evaluate(self, x, y):
        with torch.no_grad():
            act_vec = fc_forward(self.p, x, train=False, residual=self.residual)
            return _train_metrics(act_vec, y, self.current_posterior, self.regression, self.log_std)

    def predict(self, x, depth=None, get_std=False, return_model_std=False):
        with torch.no_grad():
            act_vec = fc_forward(self.p, x, depth=depth, train=False, residual=self.residual)
            if get_std:
                mu, mstd = marginalise_predict(act_vec, self.current_posterior, softmax=not self.regression,
                                               get_std=True)
                if return_model_std:
                    return mu, mstd
                return mu, (mstd ** 2 + self.log_std.exp() ** 2).pow(0.5)
            return marginalise_predict(act_vec, self.current_posterior, softmax=not self.regression)

    def exact_d_posterior(self, batches, train_bn=False):
        with torch.no_grad():
            prior = self.prior_loglikes()
            cum = 0
            for x, y in batches:
                act_vec = fc_forward(self.p, x, train=train_bn, residual=self.residual)
                joint = joint_loglike(prior, act_vec, y, self.regression, self.N, self.log_std)
                cum = cum + (x.shape[0] / self.N) * joint
            lm = marg_loglike(cum, self.logprior)
            return depth_log_posterior(cum, self.logprior, lm).exp(), lm.item()

    def exact_elbo(self, batches, train_bn=False):
        with torch.no_grad():
            prior = self.prior_loglikes()
            cum = 0.0
            for x, y in batches:
                act_vec = fc_forward(self.p, x, train=train_bn, residual=self.residual)
                e = elbo(prior, act_vec, y, self.regression, self.N, self.q_logits, self.logprior, self.log_std)
                cum = cum + (x.shape[0] / self.N) * e
            return float(cum)


# --------------------------------------------------------------------------------------------------
# synthetic inputs shared by tests and bench
# --------------------------------------------------------------------------------------------------
def wiggle_data():
    """The reference's toy 'wiggle' generator (src/datasets/additional_gap_loader.py:226-250) restated."""
    import numpy as np
    rs = np.random.RandomState(0)
    x = rs.randn(300) * 2.5 + 5
    y = np.sin(np.pi * x) + 0.2 * np.cos(np.pi * x * 4) - 0.3 * x
    y = y + rs.randn(300) * 0.25
    x, y = x[:, None], y[:, None]
    X = ((x - x.mean(axis=0)) / x.std(axis=0)).astype(np.float32)
    Y = ((y - y.mean(axis=0)) / y.std(axis=0)).astype(np.float32)
    return X, Y


def synthetic_regression(B, din, seed, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, din, generator=g, dtype=dtype)
    w = torch.randn(din, 1, generator=g, dtype=dtype) / math.sqrt(din)
    y = torch.tanh(x @ w) + 0.1 * torch.randn(B, 1, generator=g, dtype=dtype)
    return x, y
