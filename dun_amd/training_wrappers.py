"""DUN / DUN_VI training wrappers with the reference's API (src/DUN/training_wrappers.py:11-271).

Same constructor arguments, methods, return types and tensor shapes.  What differs is underneath:
  * ``fit`` runs trunk forward -> ONE fused likelihood/ELBO pass that also yields d loss / d act_vec -> trunk
    backward -> ONE fused SGD kernel over the flat parameter buffer -> ONE fused metrics kernel, and brings the
    three returned scalars to the host with a single copy (the reference syncs three times);
  * the train metrics are computed, like in the reference, from the pre-step activations with the POST-step
    q(d) for VI (training_wrappers.py:229-242) and with the pre-step exact posterior for MLL (:72-79);
  * with ``dun_amd.distributed.shard(net)`` every call works on this rank's rows of a global batch.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import kernels
from .optim import FusedSGD
from .probability import depth_categorical, homo_Gauss_mloglike
from .utils import BaseNet, cprint, to_variable


class DUN(BaseNet):
    def __init__(self, model, prob_model, N_train, lr=1e-2, momentum=0.5, weight_decay=0,
                 cuda=True, schedule=None, regression=False, pred_sig=None):
        super(DUN, self).__init__()
        cprint('y', 'DUN learnt with marginal likelihood categorical output')
        self.lr = lr
        self.momentum = momentum
        self.weight_decay = weight_decay
        self.model = model
        self.prob_model = prob_model
        self.cuda = cuda
        self.regression = regression
        self.pred_sig = pred_sig
        if self.regression:
            self.f_neg_loglike = homo_Gauss_mloglike(self.model.output_dim, self.pred_sig)
            self.f_neg_loglike_test = self.f_neg_loglike
        else:
            self.f_neg_loglike = nn.CrossEntropyLoss(reduction='none')  # takes logits
            self.f_neg_loglike_test = nn.NLLLoss(reduction='none')  # takes log probs
        self.N_train = N_train
        self._dist = None
        self.create_net()
        self.create_opt()
        self.schedule = schedule
        if self.schedule is not None and len(self.schedule) > 0:
            self.make_scheduler(gamma=0.1, milestones=self.schedule)
        self.epoch = 0

    def create_net(self):
        if self.cuda:
            self.model.cuda()
            self.f_neg_loglike.cuda()
            # The reference sets cudnn.benchmark = True here (src/DUN/training_wrappers.py:45): on cuDNN a cheap per-shape
            # algorithm pick.  On MIOpen the same flag asks for the EXHAUSTIVE search -- the first step of the 12-exit
            # ResNet-50 at batch 4096 had not finished after 18 minutes on a cold box -- while with the flag off torch's
            # MIOpen path still runs MIOpen's find in its default (hybrid) mode, which measures the applicable solvers once
            # per shape (about four minutes for that net on a cold box, seconds for the MNIST one, cached per user
            # afterwards) and lands on the same kernels.  So the flag is left as the caller has it.
        print('    Total params: %.2fM' % (self.get_nb_parameters() / 1000000.0))

    def create_opt(self):
        param_list = list(self.model.parameters()) + list(self.prob_model.parameters())
        if self.regression and self.pred_sig is None:
            param_list += list(self.f_neg_loglike.parameters())
        self.optimizer = FusedSGD(param_list, lr=self.lr, momentum=self.momentum, weight_decay=self.weight_decay)

    # ---- shared pieces of fit -----------------------------------------------------------------------
    def _prepare(self, x, y):
        x, y = to_variable(var=(x, y), cuda=self.cuda)
        if not self.regression:
            y = y.long()
        return x, y

    def _backward(self, act_vec, prior_loglikes, out):
        """Backward of loss = -objective / N_train from the fused pass's d loss / d act_vec."""
        torch.autograd.backward([act_vec], [out.dact.view_as(act_vec)])
        # the gradients of q_logits / log_std are views into this step's result block (a fresh tensor per eager step,
        # the static one of a captured step): no copy launches
        q_logits = getattr(self.prob_model, "q_logits", None)
        if q_logits is not None and q_logits.requires_grad:
            q_logits.grad = out.dq_logits
        if self.regression and self.f_neg_loglike.log_std.requires_grad:
            self.f_neg_loglike.log_std.grad = out.dlog_std.reshape(self.f_neg_loglike.log_std.shape)
        if self._dist is not None:
            self._reduce_loose_gradients()
            self._dist.wait_gradients()
        if isinstance(prior_loglikes, torch.Tensor) and prior_loglikes.requires_grad:
            # Weight-prior term of the objective (src/DUN/stochastic_fc_models.py:31-50 feeds src/probability.py:173):
            # joint_d = ... + prior_d, so d objective / d prior_d = w_d (q for the ELBO, the exact posterior for the
            # marginal); the reference's prior is the scalar sum over layers and broadcasts over depths.  It is a
            # function of the replicated parameters only, hence identical on every rank of a sharded job: its gradient
            # is accumulated into the flat gradient buffer AFTER the data-term gradients were summed over ranks (the
            # exchange would otherwise count it world-size times, and would race with this in-place accumulation).
            w = out.weights.detach()
            per_depth = prior_loglikes.dim() == 1 and prior_loglikes.shape[0] == out.D
            prior_term = (w * prior_loglikes).sum() if per_depth else w.sum() * prior_loglikes.sum()
            torch.autograd.backward([-prior_term / self.N_train])

    def _reduce_loose_gradients(self):
        """Sharded job: sum the gradients that do not live in a flat buffer (conv trunks of image DUNs)."""
        loose = [p.grad for p in self.model.parameters()
                 if getattr(p, "_dun_flat_owner", None) is None and p.grad is not None]
        if loose and self._dist.world > 1:
            flat = torch._utils._flatten_dense_tensors(loose)
            self._dist.all_reduce(flat)
            for g, r in zip(loose, torch._utils._unflatten_dense_tensors(flat, loose)):
                g.copy_(r)

    def _label_gate(self, y, act_vec):
        """Classification: the count of labels nn.CrossEntropyLoss would raise on (outside [0, C), not -100), on the
        device; the optimiser's launches leave parameters and momentum alone when it is positive, and `_metrics_host`
        raises the IndexError once the step's host copy is home -- so a caller that catches it, or a checkpoint written
        afterwards, holds the parameters from before the step, as with the reference (which raises before its update)."""
        if self.regression:
            return None
        gate_fn = getattr(kernels.backend(), "label_gate", None)
        if gate_fn is None:
            return None
        gate = gate_fn(y, act_vec.shape[-1])
        if self._dist is not None:
            self._dist.all_reduce(gate)     # one rank's bad label stops every replica's update
        return gate

    def _metrics_device(self, act_vec, y, objective, q_logits=None):
        """Device side of the returned scalars: ([objective,] nll_sum, err_sum) as one fp64 tensor + the divisor.
        ``q_logits`` (DUN_VI.fit): the metrics use q = softmax(q_logits), which also becomes ``current_posterior``."""
        be = kernels.backend()
        act = act_vec.detach()
        if q_logits is not None:
            # regression on one device: predictor, metric sums, q(d) and the objective's copy in ONE launch
            step = getattr(be, "predict_step", None)
            if step is not None and self.regression and self._dist is None and objective is not None:
                # (a captured step hands over pinned host memory: the kernel's three doubles go home without a copy node)
                pinned = self.__dict__.get("_metrics_out")
                out = pinned[:4] if pinned is not None else torch.empty(3, device=act.device, dtype=torch.float64)
                q = step(act, q_logits.detach(), y.to(torch.float32).contiguous(), self.f_neg_loglike.log_std.detach(),
                         objective.detach().reshape(1), out)
                if q is not None:
                    self.prob_model.current_posterior = q
                    if pinned is not None:
                        self._metrics_signalled = True  # out[3] is the completion word of this (captured) step
                    return out[:3], act[0].numel()
            self.prob_model.current_posterior = self.prob_model.get_q_probs().detach()
        post = self.prob_model.current_posterior
        # [objective,] nll_sum, err_sum[, bad_labels] in ONE fp64 tensor: the kernel writes its sums straight into it
        # (classification carries a third sum: the number of labels outside [0,C) other than the ignored -100)
        k = 2 if self.regression else 3
        out = torch.empty(k + (1 if objective is not None else 0), device=act.device, dtype=torch.float64)
        m = out[-k:]
        if self.regression:
            be.predict(act, post, True, get_std=True, y=y.to(torch.float32).contiguous(),
                       log_std=self.f_neg_loglike.log_std.detach(), metrics_out=m)
            n = act[0].numel()
        else:
            be.predict(act, post, False, y=y.contiguous(), metrics_out=m)
            n = act.shape[1]
        if self._dist is not None:
            self._dist.all_reduce(m)
            n = self._dist.global_batch(n)
        if objective is not None:
            out[0:1].copy_(objective.detach().reshape(1))  # float -> double inside the copy
        return out, n

    def _metrics_host(self, m, n):
        """ONE device->host copy for everything fit / eval return."""
        vals = m.tolist()
        k = 2 if self.regression else 3
        obj = vals[0] if len(vals) == k + 1 else None
        nll_sum, err_sum = vals[-k], vals[-k + 1]
        if not self.regression and vals[-1] > 0:
            # nn.CrossEntropyLoss / nn.NLLLoss raise on such labels before anything is computed (IndexError "Target ... is
            # out of bounds" on CPU); here the count comes home with the step's one host copy, so the step has run (on
            # the valid rows only: the kernels give such rows no gradient) when this is raised
            raise IndexError("Target out of bounds: %d label(s) outside [0, C) (and not the ignored -100)" % int(vals[-1]))
        nll = nll_sum / n
        err = (err_sum / n) ** 0.5 if self.regression else err_sum / n
        return obj, nll, err

    def _metrics(self, act_vec, y, objective):
        return self._metrics_host(*self._metrics_device(act_vec, y, objective))

    # ---- whole-step hipGraph for launch-bound configurations --------------------------------------------
    # A DUN_VI.fit on the UCI-sized nets is ~120 kernel launches of a few microseconds each: the step is bound by
    # launch and Python overhead, not by the GPU.  After one eager step (which also creates every lazily built
    # buffer and optimizer state) the identical launch sequence is captured once into a hipGraph and replayed;
    # inputs are copied into static tensors, the three returned scalars come back through one static tensor.
    # Re-captured whenever anything baked into the launch arguments changes (shapes, lr / momentum / weight decay,
    # frozen q_logits, re-flattened parameters).  Numerics are bit-identical to the eager step: same kernels,
    # same order.  Disabled for sharded jobs, image DUNs (MIOpen workspaces) and with DUN_NO_GRAPH=1.
    graph_steps = True
    graph_max_flop = 5e10

    def _graph_key(self, x, y):
        if not (self.graph_steps and self.cuda and self._dist is None and os.environ.get("DUN_NO_GRAPH") != "1"):
            return None
        model = self.model
        if getattr(model, "flat_params", None) is None or getattr(model, "w_prior", None) is not None:
            return None
        if not (x.is_cuda and y.is_cuda) or not getattr(model, "_all_params_flat", False):
            return None
        if 6.0 * x.shape[0] * model.flat_params.numel() > self.graph_max_flop:
            return None  # GPU-bound step: launches already hide under the kernels, nothing to gain
        g = self.optimizer.param_groups[0]
        q = getattr(self.prob_model, "q_logits", None)
        return (tuple(x.shape), x.dtype, tuple(y.shape), y.dtype, g["lr"], g["momentum"], g["weight_decay"],
                None if q is None else bool(q.requires_grad), model.flat_params.data_ptr(), getattr(model, "_flat_generation", 0),
                id(self.optimizer), float(self.N_train), type(self).__name__)

    graph_cache_size = 4  # captured steps kept per wrapper (full batch, ragged tail batch, a second loader, ...)

    def _fit_maybe_graphed(self, x, y, step_fn):
        """step_fn(x, y) -> (metrics tensor, divisor) enqueues one optimisation step without host syncs.

        One captured graph per configuration key, a few kept at a time: a loader whose last batch is ragged alternates
        between two keys every epoch and replays both instead of re-capturing.  ``prob_model.current_posterior`` after
        a replayed step is that graph's static output tensor, overwritten by its next replay (the reference hands out
        a fresh tensor per step): callers that keep it across steps clone it."""
        key = self._graph_key(x, y)
        if key is None:
            return self._metrics_host(*step_fn(x, y))
        states = self.__dict__.setdefault("_graph_states", {})
        st = states.get(key)
        if st is None:
            # first step with this configuration runs eagerly (builds buffers / optimizer state, loads kernels)
            while len(states) >= self.graph_cache_size:
                states.pop(next(iter(states)))  # oldest first
            st = states[key] = dict(key=key, graph=None)
            self._graph_state = st
            return self._metrics_host(*step_fn(x, y))
        states[key] = states.pop(key)  # most recently used last
        self._graph_state = st
        x, y = x.contiguous(), y.contiguous()
        if st["graph"] is None:
            # Second step with the same configuration: capture.  Host traffic of a replayed step is kept to memory
            # writes: the graph's first node copies x and y from wherever a two-pointer table in pinned memory says (the
            # host rewrites the table, no copy launch), and the returned scalars are written by the last kernel straight
            # into pinned memory together with a completion word the host polls (no stream synchronisation); metric
            # kernels that cannot do that leave through a device -> pinned copy node and a stream wait.
            be = kernels.backend()
            sx, sy = x.clone(), y.clone()
            graph = torch.cuda.CUDAGraph()
            pinned = torch.zeros(8, dtype=torch.float64, pin_memory=True)  # (allocated outside the capture)
            table = torch.zeros(2, dtype=torch.int64, pin_memory=True)
            indirect = getattr(be, "copy_pair_indirect", None)
            torch.cuda.synchronize()
            self._metrics_out = pinned
            try:
                with torch.cuda.graph(graph):
                    if indirect is not None:
                        indirect(sx, sy, table)
                    m, n = step_fn(sx, sy)
                    host = m
                    if m.is_cuda:  # (the metric kernels of this configuration write device memory)
                        host = pinned[:m.numel()]
                        host.copy_(m, non_blocking=True)
            finally:
                self._metrics_out = None
            # the graph holds raw pointers into the trunk's activation buffers: keep them alive as long as the graph
            # (the model's buffer cache evicts on its own schedule)
            keep = list(getattr(self.model, "_fc_cache", {}).values())
            st.update(graph=graph, x=sx, y=sy, m=m, n=n, host=host, posterior=self.prob_model.current_posterior, keep=keep,
                      table=table.numpy() if indirect is not None else None,
                      poll=pinned.numpy() if (not m.is_cuda and self.__dict__.pop("_metrics_signalled", False)) else None)
        tbl = st["table"]
        if tbl is not None:
            tbl[0], tbl[1] = x.data_ptr(), y.data_ptr()  # (x and y stay alive until this call has its results)
        else:
            st["x"].copy_(x, non_blocking=True)
            st["y"].copy_(y, non_blocking=True)
        poll = st["poll"]
        if poll is not None:
            poll[3] = 0.0
        st["graph"].replay()
        self.prob_model.current_posterior = st["posterior"]
        if poll is not None:
            for _ in range(200000):  # the step's last kernel stores 1.0 behind its scalars
                if poll[3] == 1.0:
                    break
            else:
                torch.cuda.current_stream().synchronize()
        else:
            torch.cuda.current_stream().synchronize()
        return self._metrics_host(st["host"], st["n"])

    def _prior_loglikes(self):
        """model.get_w_prior_loglike(k=None); None (= the zeros the reference returns, src/DUN/stochastic_fc_models.py:52)
        when the model declares no weight prior, which saves the fill launch of a fresh zeros tensor per step."""
        if getattr(self.model, "w_prior", 0) is None:
            return None
        return self.model.get_w_prior_loglike(k=None)

    def _forward_with_objective(self, x, y, want_grad=True, finalize=False):
        """model.forward(x), with the likelihood pass folded into the trunk's head kernels where that applies
        (regression on an FC trunk): -> (act_vec, HeadObjective or None).  ``finalize`` (a training step of a net
        without a weight prior): a trunk whose forward is one launch may finish the objective there as well."""
        spec = None
        if self.regression and getattr(self.model, "supports_head_objective", False):
            if finalize and getattr(self.model, "w_prior", 0) is None:
                spec = self.prob_model.head_objective(y, self.f_neg_loglike, x.shape[0], want_grad, N_train=self.N_train)
            else:
                spec = self.prob_model.head_objective(y, self.f_neg_loglike, x.shape[0], want_grad)
        if spec is None:
            return self.model.forward(x), None
        return self.model.forward(x, objective=spec), spec

    def _step(self, x, y):
        self.optimizer.zero_grad()
        act_vec, head_pass = self._forward_with_objective(x, y, finalize=True)
        prior_loglikes = self._prior_loglikes()
        out = self.prob_model.fused_objective(prior_loglikes, act_vec, y, self.f_neg_loglike, self.N_train,
                                              head_pass=head_pass)
        self._backward(act_vec, prior_loglikes, out)
        self.optimizer.step(gate=self._label_gate(y, act_vec))
        # this posterior is one iteration behind the parameters: estimated with the acts from before the step
        self.prob_model.current_posterior = out.posterior.clone()
        return self._metrics_device(act_vec, y, out.log_marg)

    def fit(self, x, y):
        """Optimise the stochastically estimated marginal joint of parameters and weights"""
        self.model.train()
        x, y = self._prepare(x, y)
        return self._fit_maybe_graphed(x, y, self._step)

    def eval(self, x, y):
        self.model.eval()
        with torch.no_grad():
            x, y = self._prepare(x, y)
            act_vec = self.model.forward(x)
            _, nll, err = self._metrics(act_vec, y, None)
            return nll, err

    def layer_predict(self, x):
        self.model.eval()
        x, = to_variable(var=(x, ), cuda=self.cuda)
        with torch.no_grad():
            out = self.model.forward(x).data
        if not self.regression:
            out = F.softmax(out, dim=2)
        return out

    def vec_predict(self, x, bin_mat):
        """Get predictions for specific binary vector configurations (cold path: torch ops)"""
        self.model.eval()
        x, = to_variable(var=(x, ), cuda=self.cuda)
        out = x.data.new(bin_mat.shape[0], x.shape[0], self.model.output_dim)
        with torch.no_grad():
            for s in range(bin_mat.shape[0]):
                out[s] = self.model.vec_forward(x, bin_mat[s, :]).data
        # (the reference only defines `probs` for classification and raises for regression)
        if not self.regression:
            probs = F.softmax(out, dim=2)
        return probs.data

    def predict(self, x, depth=None, get_std=False, return_model_std=False):
        self.model.eval()
        with torch.no_grad():
            x, = to_variable(var=(x,), cuda=self.cuda)
            act_vec = self.model.forward(x, depth=depth).data
            if get_std:
                # (model_std ** 2 + log_std.exp() ** 2).pow(0.5) of the reference (src/DUN/training_wrappers.py:153) is
                # taken inside the predictor's launch: five element-wise launches fewer per call
                noise = None if (return_model_std or not self.regression) else self.f_neg_loglike.log_std
                fused = noise is not None and noise.numel() == 1
                pred_mu, std = depth_categorical.marginalise_d_predict(
                    act_vec, self.prob_model.current_posterior, depth=depth, softmax=(not self.regression),
                    get_std=get_std, noise_log_std=noise if fused else None)
                if noise is not None and not fused:
                    std = (std ** 2 + noise.exp() ** 2).pow(0.5)
                return pred_mu.data, std.data
            probs = depth_categorical.marginalise_d_predict(act_vec, self.prob_model.current_posterior, depth=depth,
                                                            softmax=(not self.regression), get_std=get_std)
            return probs.data

    def fast_predict(self, x):
        self.model.eval()
        with torch.no_grad():
            x, = to_variable(var=(x,), cuda=self.cuda)
            act_vec = self.model.fast_forward_impl(x, self.prob_model.current_posterior, min_prob=1e-2).data
            probs = depth_categorical.marginalise_d_predict(act_vec, self.prob_model.current_posterior, depth=None,
                                                            softmax=True, get_std=False)
            return probs.data

    def _sweep(self, trainloader, train_bn, vi):
        if train_bn:
            self.model.train()
        else:
            self.model.eval()
        with torch.no_grad():
            prior_loglikes = self.model.get_w_prior_loglike(k=None)
            N_train = len(trainloader.dataset)
            world = 1 if self._dist is None else self._dist.world
            # sharded: every rank's loader yields its rows of each global batch (the dataset it wraps is this rank's
            # 1/world of the data, or the whole set behind a DistributedSampler); the objective of a batch is already
            # the all-reduced, global-batch one
            assert N_train == self.N_train or N_train * world == self.N_train
            N_train = self.N_train
            cum = None
            for x, y in trainloader:
                x, y = self._prepare(x, y)
                act_vec, head_pass = self._forward_with_objective(x, y, want_grad=False)
                gb = x.shape[0] * world
                if vi:
                    out = self.prob_model.fused_objective(prior_loglikes, act_vec, y, self.f_neg_loglike, N_train,
                                                          want_grad=False, head_pass=head_pass)
                    term = (gb / N_train) * out.objective.double()
                else:
                    out = depth_categorical.fused_objective(self.prob_model, prior_loglikes, act_vec, y,
                                                            self.f_neg_loglike, N_train, want_grad=False,
                                                            head_pass=head_pass)
                    term = (gb / N_train) * out.joint.double()
                cum = term if cum is None else cum + term
            return cum

    def get_exact_d_posterior(self, trainloader, train_bn=False, logposterior=False):
        """Exact posterior over depth and log marginal over weights with a full pass over the data"""
        cum = self._sweep(trainloader, train_bn, vi=False).float()
        log_marg = self.prob_model.get_marg_loglike(cum)
        log_post = self.prob_model.get_depth_log_posterior(cum, log_marg)
        exact_posterior = log_post if logposterior else log_post.exp()
        return exact_posterior, log_marg.item()


class DUN_VI(DUN):
    def __init__(self, model, prob_model, N_train, lr=1e-2, momentum=0.5, weight_decay=0, cuda=True,
                 schedule=None, regression=False, pred_sig=None):
        super(DUN_VI, self).__init__(model, prob_model, N_train, lr, momentum, weight_decay, cuda,
                                     schedule, regression, pred_sig)

    def _step(self, x, y):
        self.optimizer.zero_grad()
        act_vec, head_pass = self._forward_with_objective(x, y, finalize=True)
        prior_loglikes = self._prior_loglikes()
        out = self.prob_model.fused_objective(prior_loglikes, act_vec, y, self.f_neg_loglike, self.N_train, Beta=1,
                                              head_pass=head_pass)
        self._backward(act_vec, prior_loglikes, out)
        self.optimizer.step(gate=self._label_gate(y, act_vec))
        # train metrics with the POST-step q(d), which also becomes current_posterior (training_wrappers.py:229-242)
        return self._metrics_device(act_vec, y, out.objective, q_logits=self.prob_model.q_logits)

    def fit(self, x, y):
        """Optimise the ELBO over weights and q(d)"""
        self.set_mode_train(train=True)
        x, y = self._prepare(x, y)
        return self._fit_maybe_graphed(x, y, self._step)

    def get_exact_ELBO(self, trainloader, train_bn=False):
        """Exact ELBO with a full pass over the data"""
        return self._sweep(trainloader, train_bn, vi=True).item()
