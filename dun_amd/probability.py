"""Depth-categorical probability module with the reference's class and method surface.

Reference: src/probability.py:46-60 (homo_Gauss_mloglike), :136-208 (depth_categorical), :211-257
(depth_categorical_VI).  Per-activation likelihoods, per-depth reductions, the logsumexp marginal, the
categorical KL, the ELBO, the depth-marginal predictor and the backward of all of them run in the fused HIP
kernels of dun_amd/csrc/elbo.hip; the D-vector bookkeeping that remains in torch is a handful of elements.

Besides the reference methods there is ``fused_objective``: one call that returns the objective together
with d(-objective/N)/d(act, q_logits, log_std) -- for the VI objective the backward needs no reduced
quantity, so forward and backward are a single streaming pass over ``act_vec``.  The training wrappers use it;
the reference-named methods stay differentiable through autograd for any other caller.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import kernels
from ._lib import OBJ_MLL, OBJ_VI, RES_DLOG_STD, RES_E_LOGLIKE, RES_HEADER, RES_KL, RES_LOG_MARG, RES_OBJECTIVE


class diag_w_Gauss_loglike(object):
    """Factorised Gaussian weight prior of the reference's regression experiments (src/probability.py:30-40):
    ``prior(w)`` = multiplier * sum log N(w; mu, scale).  As in the reference the second argument, although named
    sigma^2, is used as the SCALE (standard deviation) of the Normal.  An O(#parameters) torch reduction evaluated
    once per step on the replicated weights -- not part of the per-sample path."""

    def __init__(self, μ=0, σ2=1, multiplier=1):
        self.μ, self.σ2, self.multiplier = μ, σ2, multiplier
        self._const = -float(np.log(σ2)) - 0.5 * float(np.log(2 * np.pi))

    def __call__(self, w):
        return (self._const * w.numel() - ((w - self.μ) ** 2).sum() / (2 * self.σ2 ** 2)) * self.multiplier

    def summary(self):
        return {"name": "diag_w_Gauss", "μ": self.μ, "σ2": self.σ2, "mul": self.multiplier}


class homo_Gauss_mloglike(nn.Module):
    """Homoscedastic Gaussian negative log-likelihood with a learnable (or fixed) log std."""

    def __init__(self, Ndims=1, sig=None):
        super(homo_Gauss_mloglike, self).__init__()
        if sig is None:
            self.log_std = nn.Parameter(torch.zeros(Ndims))
        else:
            self.log_std = nn.Parameter(torch.ones(Ndims) * np.log(sig), requires_grad=False)

    def forward(self, mu, y, model_std=None):
        # Elementwise helper kept for callers outside the fused path (e.g. evaluation scripts); the training
        # and prediction wrappers never call it: their NLL is fused into the kernels.
        sig = self.log_std.exp().clamp(min=1e-4)
        if model_std is not None:
            sig = (sig ** 2 + model_std ** 2) ** 0.5
        return 0.5 * np.log(2 * np.pi) + torch.log(sig) + (y - mu) ** 2 / (2 * sig ** 2)


def _likelihood_kind(f_neg_loglike):
    """(regression?, log_std or None) of a likelihood object the kernels implement."""
    if isinstance(f_neg_loglike, homo_Gauss_mloglike):
        return True, f_neg_loglike.log_std
    if isinstance(f_neg_loglike, nn.CrossEntropyLoss):
        return False, None
    raise TypeError("dun_amd implements homo_Gauss_mloglike and nn.CrossEntropyLoss(reduction='none') "
                    "likelihoods; got %r" % type(f_neg_loglike))


def _prior_vec(prior_loglikes, D, ref):
    """The reference adds prior_loglikes by broadcasting: a [D] vector, or the scalar the w_prior path returns."""
    if prior_loglikes is None:
        return None
    p = prior_loglikes.detach().to(device=ref.device, dtype=torch.float32)
    if p.dim() == 0 or p.numel() == 1:
        p = p.reshape(1).expand(D)
    return p.contiguous()


class HeadObjective(object):
    """Request to a trunk's forward: run the Gaussian likelihood pass inside the head kernels (the head value of a row
    is in a register when it is written, so y - act, d loss / d act and the per-depth sums of squared residuals cost
    no further pass over act_vec).  The forward fills ``partials`` (double [D, nparts]) and, when depth weights were
    given, ``dact`` ([D,B,1]); ``_fused(..., head_pass=spec)`` then skips its own streaming pass."""
    __slots__ = ("y", "w", "w_is_logits", "log_std", "inv_batch", "partials", "dact", "fin", "res", "sums")

    def __init__(self, y, w, w_is_logits, log_std, inv_batch, fin=None):
        self.y, self.w, self.w_is_logits, self.log_std, self.inv_batch = y, w, w_is_logits, log_std, inv_batch
        self.partials = self.dact = None
        # fin: the depth-level arguments of the objective (dict: objective, q_logits, logprior, N_train, batch, beta, eps).
        # A trunk whose forward is one launch finishes the objective there and leaves the result block in `res` (and the
        # per-depth sums in `sums`); otherwise both stay None and _fused launches the finalisation itself.
        self.fin, self.res, self.sums = fin, None, None


class FusedOut(object):
    """Result of depth_categorical.fused_objective (all tensors stay on the device)."""
    __slots__ = ("res", "D", "dact", "sums")

    def __init__(self, res, D, dact, sums):
        self.res, self.D, self.dact, self.sums = res, D, dact, sums

    def _arr(self, k):
        return self.res[RES_HEADER + k * self.D: RES_HEADER + (k + 1) * self.D]

    objective = property(lambda s: s.res[RES_OBJECTIVE])
    E_loglike = property(lambda s: s.res[RES_E_LOGLIKE])
    KL = property(lambda s: s.res[RES_KL])
    log_marg = property(lambda s: s.res[RES_LOG_MARG])
    dlog_std = property(lambda s: s.res[RES_DLOG_STD:RES_DLOG_STD + 1])
    joint = property(lambda s: s._arr(0))
    weights = property(lambda s: s._arr(1))      # q (VI) or exact posterior (MLL)
    dq_logits = property(lambda s: s._arr(2))
    posterior = property(lambda s: s._arr(3))    # exact posterior, always
    loglike_sum = property(lambda s: s._arr(4))  # sum_b loglike[d,b]


def head_objective_spec(y, f_neg_loglike, objective, q_logits, batch, dist=None, want_grad=True, finalize=None):
    """A HeadObjective for this step, or None where the fusion does not apply (classification; more than one output).
    ``finalize`` (dict: logprior, N_train, beta, eps; single-device jobs without a weight prior) also asks the forward
    to finish the objective where it can."""
    if not isinstance(f_neg_loglike, homo_Gauss_mloglike) or f_neg_loglike.log_std.numel() != 1:
        return None
    if y.dim() > 2 or (y.dim() == 2 and y.shape[1] != 1) or y.shape[0] != batch:
        return None
    gb = batch if dist is None else dist.global_batch(batch)
    yy = y.detach().to(torch.float32).contiguous()
    w = q_logits.detach() if (objective == OBJ_VI and want_grad) else None
    fin = None
    if finalize is not None and dist is None:
        fin = dict(objective=objective, q_logits=None if q_logits is None else q_logits.detach(),
                   logprior=finalize["logprior"].to(device=yy.device, dtype=torch.float32).contiguous(),
                   N_train=float(finalize["N_train"]), batch=float(gb), beta=float(finalize.get("beta", 1.0)),
                   eps=float(finalize.get("eps", 1e-35)),
                   # the ELBO's d loss / d act needs no reduced quantity: its finalisation may wait for the backward's
                   # weight-gradient launch; the exact marginal needs the posterior before its backward
                   defer=bool(objective == OBJ_VI and want_grad))
    return HeadObjective(yy, w, True, f_neg_loglike.log_std.detach(), 1.0 / gb, fin)


def _fused(act_vec, y, f_neg_loglike, N_train, logprior, prior_loglikes, objective, q_logits=None, beta=1.0,
           eps=1e-35, want_grad=True, dist=None, want_ll=False, head_pass=None):
    be = kernels.backend()
    regression, log_std = _likelihood_kind(f_neg_loglike)
    act = act_vec.detach()
    D, B = act.shape[0], act.shape[1]
    if not regression and act.dim() != 3:
        raise RuntimeError("classification expects act_vec [D,B,C]")
    act = act.to(torch.float32).contiguous()
    yy = y.detach().contiguous()
    if regression:
        yy = yy.to(torch.float32)
    ls = None if log_std is None else log_std.detach()
    gb = B if dist is None else dist.global_batch(B)
    logprior = logprior.to(device=act.device, dtype=torch.float32)
    prior = _prior_vec(prior_loglikes, D, act)
    ql = None if q_logits is None else q_logits.detach()
    ll = None
    # single-device jobs have nothing to exchange between the per-workgroup partial sums and the finalisation: the
    # partials are added up inside the finalisation launch, so the whole objective is two launches
    local = dist is None

    # the trunk's head kernels already ran the pass (HeadObjective): its partial sums (and dact) are used as they are
    from_head = head_pass is not None and head_pass.partials is not None and regression and not want_ll
    # ... and, where the whole forward was one launch, the depth-level finalisation too (head_pass.res): taken only if it
    # was computed from the very arguments of this call
    done = None
    if from_head and head_pass.res is not None and local and prior is None:
        f = head_pass.fin
        same_q = ql is None or (f["q_logits"] is not None and f["q_logits"].data_ptr() == ql.data_ptr())
        if (f["objective"] == objective and f["N_train"] == float(N_train) and f["beta"] == float(beta)
                and f["eps"] == float(eps) and f["batch"] == float(gb) and same_q
                and f["logprior"].data_ptr() == logprior.contiguous().data_ptr()):
            done = head_pass

    def finalize(s_, q_, obj):
        if done is not None:
            return done.res, done.sums
        if local:
            return be.depth_finalize(s_, regression, q_, logprior, prior, ls, N_train, gb, beta, eps, obj, partials=True)
        dist.all_reduce(s_)
        return be.depth_finalize(s_, regression, q_, logprior, prior, ls, N_train, gb, beta, eps, obj), s_

    def head_sums():
        return head_pass.partials if local else be.ll_reduce(head_pass.partials)

    if objective == OBJ_VI:
        # q is known before the pass: likelihood, its reduction and dact in ONE sweep over act_vec
        if from_head and (head_pass.dact is not None or not want_grad):
            sums, dact = head_sums(), (head_pass.dact if want_grad else None)
        else:
            sums, dact, ll = be.ll_pass(act, yy, regression, ls, ql if want_grad else None, True, 1.0 / gb,
                                        want_dact=want_grad, want_ll=want_ll, reduce=not local)
        res, sums = finalize(sums, ql, OBJ_VI)
    else:
        # the exact posterior needs the complete (all-reduced) per-depth sums first: two sweeps
        if from_head:
            sums = head_sums()
        else:
            sums, _, ll = be.ll_pass(act, yy, regression, ls, None, False, 1.0 / gb, want_dact=False, want_ll=want_ll,
                                     reduce=not local)
        res, sums = finalize(sums, None, OBJ_MLL)
        dact = None
        if want_grad:
            post = res[RES_HEADER + 3 * D: RES_HEADER + 4 * D]
            _, dact, _ = be.ll_pass(act, yy, regression, ls, post, False, 1.0 / gb, want_dact=True)
    out = FusedOut(res, D, dact, sums)
    if want_ll:
        return out, ll
    return out


def _autograd_route_is_single_device(dist, act_vec):
    """The differentiable reference-named entries (estimate_ELBO / get_w_joint_loglike + loss.backward()) leave the
    gradient exchange of a sharded job to nobody: only fit() orders trunk backward -> all-reduce -> wait -> prior
    gradient -> step.  Refuse instead of training replicas apart."""
    if dist is not None and dist.world > 1 and act_vec.requires_grad:
        raise RuntimeError("batch-sharded job: differentiate through net.fit(), not through estimate_ELBO / "
                           "get_w_joint_loglike + backward() (nothing would wait for the gradient all-reduce)")


class _JointLoglike(torch.autograd.Function):
    """joint[d] = (N/B) sum_b loglike[d,b] + prior[d], differentiable w.r.t. act_vec, log_std and the prior."""

    @staticmethod
    def forward(ctx, act_vec, log_std, prior, y, f_neg_loglike, N_train, dist):
        _autograd_route_is_single_device(dist, act_vec)
        D = act_vec.shape[0]
        out = _fused(act_vec, y, f_neg_loglike, N_train, torch.zeros(D), prior, OBJ_MLL, want_grad=False, dist=dist)
        ctx.save_for_backward(act_vec, y, out.sums)
        ctx.f, ctx.N, ctx.dist = f_neg_loglike, N_train, dist
        ctx.prior_shape = None if prior is None else prior.shape
        return out.joint.clone()

    @staticmethod
    def backward(ctx, gj):
        act_vec, y, sums = ctx.saved_tensors
        be = kernels.backend()
        regression, log_std = _likelihood_kind(ctx.f)
        B = act_vec.shape[1]
        gb = B if ctx.dist is None else ctx.dist.global_batch(B)
        gj = gj.to(torch.float32).contiguous()
        # kernel convention: dact = -(w_d / batch) * dll/dact  ->  w = -gj * N gives gj * (N/batch) * dll/dact
        w = (-gj * float(ctx.N)).contiguous()
        ls = None if log_std is None else log_std.detach()
        yy = y.to(torch.float32) if regression else y
        _, dact, _ = be.ll_pass(act_vec.detach().to(torch.float32).contiguous(), yy.contiguous(), regression, ls, w,
                                False, 1.0 / gb, want_dact=True)
        g_ls = None
        if regression and log_std.requires_grad:
            sig_raw = ls.exp()
            sig = sig_raw.clamp(min=1e-4)
            per_depth = (sums.to(torch.float32) / sig ** 2 - gb) * (float(ctx.N) / gb)
            g_ls = ((gj * per_depth).sum() * (sig_raw >= 1e-4).to(torch.float32)).reshape(log_std.shape)
        g_prior = None
        if ctx.prior_shape is not None:
            g_prior = gj if len(ctx.prior_shape) == 1 and ctx.prior_shape[0] == gj.shape[0] else gj.sum().reshape(ctx.prior_shape)
        return dact.view_as(act_vec), g_ls, g_prior, None, None, None, None


class depth_categorical(nn.Module):
    def __init__(self, prior_probs, prior_logprobs=None, cuda=True):
        super(depth_categorical, self).__init__()
        self.prior_probs = torch.Tensor(prior_probs)
        assert self.prior_probs.sum().item() - 1 < 1e-6
        self.dims = self.prior_probs.shape[0]
        if prior_logprobs is None:
            self.logprior = self.prior_probs.log()
        else:
            self.logprior = torch.Tensor(prior_logprobs)
            self.prior_probs = self.logprior.exp()
            assert self.prior_probs.sum().item() - 1 < 1e-6
        self.current_posterior = None
        self._dist = None
        self.cuda = cuda
        if self.cuda:
            self.to_cuda()

    def to_cuda(self):
        self.prior_probs = self.prior_probs.cuda()
        self.logprior = self.logprior.cuda()

    def __getstate__(self):
        st = self.__dict__.copy()
        st["_dist"] = None  # process groups do not pickle; re-shard after loading
        return st

    @staticmethod
    def get_w_joint_loglike(prior_loglikes, act_vec, y, f_neg_loglike, N_train, dist=None):
        """Note that if we average this to get the exact joint, all batches need the same size; alternatively
        weigh each component with its batch size (as get_exact_d_posterior does)."""
        regression, log_std = _likelihood_kind(f_neg_loglike)
        return _JointLoglike.apply(act_vec, log_std, prior_loglikes, y, f_neg_loglike, N_train, dist)

    def get_marg_loglike(self, joint_loglike_per_depth):
        return torch.logsumexp(joint_loglike_per_depth + self.logprior, dim=0)

    def get_depth_log_posterior(self, joint_loglike_per_depth, log_marginal_over_depth=None):
        if log_marginal_over_depth is None:
            log_marginal_over_depth = self.get_marg_loglike(joint_loglike_per_depth)
        return joint_loglike_per_depth + self.logprior - log_marginal_over_depth

    @staticmethod
    def marginalise_d_predict(act_vec, d_posterior, depth=None, softmax=False, get_std=False, noise_log_std=None):
        """Predict while marginalising d with the given distribution (fused kernel, no autograd).
        ``noise_log_std`` (not in the reference's signature; regression with get_std): the std handed back is the total
        predictive one, sqrt(model_std^2 + exp(noise_log_std)^2) -- what DUN.predict adds behind this call
        (src/DUN/training_wrappers.py:153), taken in the predictor's own launch."""
        assert not (softmax and get_std)
        act = act_vec.detach().to(torch.float32).contiguous()
        if noise_log_std is not None and get_std and not softmax:
            out, std, _ = kernels.backend().predict(act, d_posterior, regression=True, get_std=True,
                                                    log_std=noise_log_std.detach().to(torch.float32).reshape(-1))
            return out, std
        out, std, _ = kernels.backend().predict(act, d_posterior, regression=not softmax, get_std=get_std)
        if get_std:
            return out, std
        return out

    def fused_objective(self, prior_loglikes, act_vec, y, f_neg_loglike, N_train, want_grad=True, want_ll=False,
                        head_pass=None):
        """Exact marginal over depth + its backward (see module docstring).  -> FusedOut"""
        return _fused(act_vec, y, f_neg_loglike, N_train, self.logprior, prior_loglikes, OBJ_MLL,
                      want_grad=want_grad, dist=self._dist, want_ll=want_ll, head_pass=head_pass)

    def head_objective(self, y, f_neg_loglike, batch, want_grad=True, N_train=None):
        """HeadObjective for the exact-marginal objective (sums only: the posterior is not known before the pass).
        With ``N_train`` the forward may also finish the objective (no weight prior, one device)."""
        fin = None if N_train is None else dict(logprior=self.logprior, N_train=N_train)
        return head_objective_spec(y, f_neg_loglike, OBJ_MLL, None, batch, self._dist, want_grad, fin)


class _Elbo(torch.autograd.Function):
    """ELBO as one fused forward+backward pass; autograd just rescales the gradients computed in forward."""

    @staticmethod
    def forward(ctx, act_vec, q_logits, log_std, prior, y, f_neg_loglike, N_train, pm, beta):
        _autograd_route_is_single_device(pm._dist, act_vec)
        out = _fused(act_vec, y, f_neg_loglike, N_train, pm.logprior, prior, OBJ_VI, q_logits=q_logits, beta=beta,
                     eps=pm.eps, want_grad=True, dist=pm._dist)
        ctx.save_for_backward(out.dact, out.res)
        ctx.N, ctx.D = float(N_train), out.D
        ctx.ls_shape = None if log_std is None else log_std.shape
        ctx.prior_shape = None if prior is None else prior.shape
        ctx.act_shape = act_vec.shape
        return out.objective.clone()

    @staticmethod
    def backward(ctx, g):
        dact, res = ctx.saved_tensors
        D = ctx.D
        s = g * (-ctx.N)  # the kernels differentiated -ELBO/N
        g_act = (dact * s).view(ctx.act_shape)
        g_q = res[RES_HEADER + 2 * D: RES_HEADER + 3 * D] * s
        g_ls = None if ctx.ls_shape is None else (res[RES_DLOG_STD] * s).reshape(ctx.ls_shape)
        g_prior = None
        if ctx.prior_shape is not None:
            q = res[RES_HEADER + D: RES_HEADER + 2 * D]
            g_prior = g * q if len(ctx.prior_shape) == 1 and ctx.prior_shape[0] == D else (g * q.sum()).reshape(ctx.prior_shape)
        return g_act, g_q, g_ls, g_prior, None, None, None, None, None


class depth_categorical_VI(depth_categorical):
    def __init__(self, prior_probs, cuda=True, eps=1e-35):
        super(depth_categorical_VI, self).__init__(prior_probs, None, cuda)
        self.q_logits = nn.Parameter(torch.zeros(self.dims), requires_grad=True)
        self.eps = eps
        if cuda:
            self.to_cuda_VI()

    def to_cuda_VI(self):
        self.q_logits.data = self.q_logits.data.cuda()

    def get_q_logprobs(self):
        return F.log_softmax(self.q_logits, dim=0)

    def get_q_probs(self):
        return F.softmax(self.q_logits, dim=0)

    def get_KL(self):
        log_q = self.get_q_logprobs()
        q = self.get_q_probs().clamp(min=self.eps, max=(1 - self.eps))
        return (q * (log_q - self.logprior)).sum(dim=0)

    def get_E_loglike(self, joint_loglike_per_depth):
        return (self.get_q_probs() * joint_loglike_per_depth).sum(dim=0)

    def estimate_ELBO(self, prior_loglikes, act_vec, y, f_neg_loglike, N_train, Beta=1):
        """ELBO on the log joint of data and network weights (differentiable; fused kernel underneath)."""
        regression, log_std = _likelihood_kind(f_neg_loglike)
        return _Elbo.apply(act_vec, self.q_logits, log_std, prior_loglikes, y, f_neg_loglike, N_train, self, Beta)

    def fused_objective(self, prior_loglikes, act_vec, y, f_neg_loglike, N_train, want_grad=True, want_ll=False,
                        Beta=1, head_pass=None):
        """ELBO + d(-ELBO/N)/d(act, q_logits, log_std) in a single pass over act_vec.  -> FusedOut"""
        return _fused(act_vec, y, f_neg_loglike, N_train, self.logprior, prior_loglikes, OBJ_VI,
                      q_logits=self.q_logits, beta=Beta, eps=self.eps, want_grad=want_grad, dist=self._dist,
                      want_ll=want_ll, head_pass=head_pass)

    def head_objective(self, y, f_neg_loglike, batch, want_grad=True, N_train=None, Beta=1):
        """HeadObjective for the ELBO: q is known before the pass, so the head kernels also write d loss / d act.
        With ``N_train`` the forward may also finish the objective (no weight prior, one device)."""
        fin = None if N_train is None else dict(logprior=self.logprior, N_train=N_train, beta=Beta, eps=self.eps)
        return head_objective_spec(y, f_neg_loglike, OBJ_VI, self.q_logits, batch, self._dist, want_grad, fin)

    def q_predict(self, act_vec, depth=None, softmax=False):
        return depth_categorical.marginalise_d_predict(act_vec, self.get_q_probs(), depth=depth, softmax=softmax)
