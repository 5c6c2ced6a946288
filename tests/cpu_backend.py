"""TEST INFRASTRUCTURE: a CPU stand-in for dun_amd.kernels.HipBackend, written with plain torch ops.

It exists so that the host-side logic above the C ABI -- staging of the trunk at the BatchNorm exchange points,
batch sharding, gradient/likelihood-sum all-reduces, the wrappers' bookkeeping -- can be exercised by
multi-process gloo tests on machines without a GPU.  It mirrors the kernels' contracts (same staged calls, same
result-block layout), not their implementation.  Never imported by the product package.
"""
import math

import torch
import torch.nn.functional as F

H_ = 8  # result header (DUN_RES_HEADER)


class CpuFcBuffers(object):
    def __init__(self, model, B, depth, train):
        W, O, L = model.width, model.output_dim, model.n_layers
        self.model, self.B, self.depth, self.train = model, B, depth, train
        self.H = torch.zeros(depth + 1, B, W)
        self.Z = torch.zeros(depth if train else min(depth, 1), B, W)
        self.act = None
        self._act_shape = (depth + 1, B, O)
        self.bn_mean = torch.zeros(max(L, 1), W)
        self.bn_rstd = torch.zeros(max(L, 1), W)
        self.bn_sums = torch.zeros(2, W, dtype=torch.float64)
        self.G = self.dU = self.x = self.dact = None
        self._obj = None
        self.generation = 0  # bumped by every forward that writes these buffers

    def new_act(self):
        self.act = torch.zeros(self._act_shape)

    def set_input(self, x):
        self.x = x

    def ensure_backward(self):
        if self.G is None:
            self.G = torch.zeros(self.B, self.H.shape[2])
            self.dU = torch.zeros_like(self.G)

    def set_dact(self, dact):
        self.dact = dact

    def set_objective(self, spec):
        """Mirror of FcBuffers.set_objective: the heads below fill spec.partials (one partial per depth here) and,
        with depth weights, spec.dact."""
        self._obj = spec
        if spec is not None:
            D, B, O = self._act_shape
            assert O == 1
            spec.partials = torch.zeros(D, 1, dtype=torch.float64)
            spec.dact = torch.zeros(self._act_shape) if spec.w is not None else None

    @property
    def head_fused(self):
        m = self.model
        return m.output_dim == 1 and m.width % 256 == 0 and m.width <= 2048 and self.depth > 0


class CpuBackend(object):
    name = "cpu-standin"

    # ---- probability module ----
    def ll_pass(self, act, y, regression, log_std, w, w_is_logits, inv_batch, want_dact=True, want_ll=False,
                reduce=True):
        """(`reduce=False`: the kernels hand back per-workgroup partials [D, nparts]; here one partial per depth)"""
        sums, dact, ll = self._ll_pass(act, y, regression, log_std, w, w_is_logits, inv_batch, want_dact, want_ll)
        return (sums if reduce else sums.view(-1, 1)), dact, ll

    def _ll_pass(self, act, y, regression, log_std, w, w_is_logits, inv_batch, want_dact, want_ll):
        D, B = act.shape[0], act.shape[1]
        wt = None
        if w is not None:
            wt = F.softmax(w.float(), dim=0) if w_is_logits else w.float()
        if regression:
            if act.dim() == 3 and act.shape[2] != 1:
                raise RuntimeError("shape '[%d, %d]' is invalid for input of size %d" % (D, B, act.numel()))
            a = act.reshape(D, B)
            r = y.reshape(1, B) - a
            sig = log_std.exp().clamp(min=1e-4)
            sums = (r.double() ** 2).sum(dim=1)
            ll = -(0.5 * math.log(2 * math.pi) + torch.log(sig) + r ** 2 / (2 * sig ** 2)) if want_ll else None
            dact = None
            if want_dact:
                dact = (-(wt.view(D, 1) * inv_batch) * r / sig ** 2).reshape(act.shape)
            return sums, dact, ll
        # label semantics of the kernels (include/dun_hip.h, dun_ll_softmax): -100 is ignored, any other label outside
        # [0,C) poisons its depth's sum with NaN; neither gets a gradient
        C = act.shape[2]
        ok = (y >= 0) & (y < C)
        ys = torch.where(ok, y, torch.zeros_like(y))
        lsm = F.log_softmax(act, dim=2)
        llv = lsm.gather(2, ys.view(1, B, 1).expand(D, B, 1)).squeeze(2)
        fallback = torch.where(y == -100, torch.zeros(B), torch.full((B,), float("nan")))
        llv = torch.where(ok.view(1, B), llv, fallback.view(1, B).to(llv.dtype))
        sums = llv.double().sum(dim=1)
        dact = None
        if want_dact:
            oh = F.one_hot(ys, C).to(act.dtype).unsqueeze(0)
            dact = -(wt.view(D, 1, 1) * inv_batch) * (oh - lsm.exp()) * ok.view(1, B, 1).to(act.dtype)
        return sums, dact, (llv if want_ll else None)

    def ll_reduce(self, partials):
        return partials.sum(dim=1)

    def depth_finalize(self, sums, gaussian, q_logits, logprior, prior_ll, log_std, N_train, batch, beta, eps,
                       objective, partials=False):
        if partials:
            s_ = sums.sum(dim=1)
            return self.depth_finalize(s_, gaussian, q_logits, logprior, prior_ll, log_std, N_train, batch, beta, eps,
                                       objective), s_
        D = sums.shape[0]
        res = torch.zeros(H_ + 5 * D)
        lp = logprior.double()
        prior = torch.zeros(D, dtype=torch.float64) if prior_ll is None else prior_ll.double()
        live = 1.0
        if gaussian:
            raw = float(log_std.exp())
            live = 1.0 if raw >= 1e-4 else 0.0
            sig = max(raw, 1e-4)
            LL = -batch * (0.5 * math.log(2 * math.pi) + math.log(sig)) - 0.5 * sums / sig ** 2
        else:
            LL = sums
        J = (N_train / batch) * LL + prior
        log_marg = torch.logsumexp(J + lp, dim=0)
        post = (J + lp - log_marg).exp()
        wgt, dq = post, torch.zeros(D, dtype=torch.float64)
        E = KL = torch.zeros((), dtype=torch.float64)
        if objective == 0:
            q = F.softmax(q_logits.float(), dim=0).double()
            lq = F.log_softmax(q_logits.float(), dim=0).double()
            c = q.clamp(min=eps, max=1 - eps)
            m = ((q >= eps) & (q <= 1 - eps)).double()
            E, KL = (q * J).sum(), (c * (lq - lp)).sum()
            dKL = m * q * (lq - lp) - q * (m * q * (lq - lp)).sum() + c - q * c.sum()
            dq = -(1.0 / N_train) * (q * (J - E) - beta * dKL)
            wgt = q
        res[0] = E - beta * KL if objective == 0 else log_marg
        res[1], res[2], res[3] = E, KL, log_marg
        if gaussian:
            res[4] = -(1.0 / batch) * (wgt * (sums / sig ** 2 - batch)).sum() * live
        for k, v in enumerate((J, wgt, dq, post, LL)):
            res[H_ + k * D:H_ + (k + 1) * D] = v.float()
        return res

    def predict(self, act, q, regression, get_std=False, y=None, log_std=None, metrics_out=None):
        q = q.detach().float()
        qb = q.view(-1, *([1] * (act.dim() - 1)))
        if regression:
            mu = (qb * act).sum(0)
            std = ((qb * act ** 2).sum(0) - mu ** 2).pow(0.5)
            metrics = None
            if y is not None:
                sig = log_std.exp().clamp(min=1e-4)
                st = (sig ** 2 + std ** 2) ** 0.5
                nll = 0.5 * math.log(2 * math.pi) + torch.log(st) + (y - mu) ** 2 / (2 * st ** 2)
                metrics = torch.stack([nll.double().sum(), ((mu - y).double() ** 2).sum()])
                if metrics_out is not None:
                    metrics = metrics_out.copy_(metrics)
            if y is None and log_std is not None:  # total predictive std (dun_predict_gauss: log_std without targets)
                std = (std ** 2 + log_std.exp() ** 2).pow(0.5)
            return mu, (std if (get_std or y is not None) else None), metrics
        probs = (qb * F.softmax(act, dim=2)).sum(0)
        metrics = None
        if y is not None:
            C = act.shape[2]
            ok = (y >= 0) & (y < C)
            ys = torch.where(ok, y, torch.zeros_like(y))
            row = -torch.log(probs.gather(1, ys.view(-1, 1))).squeeze(1).double()
            bad = ~ok & (y != -100)
            row = torch.where(ok, row, torch.where(bad, torch.full_like(row, float("nan")), torch.zeros_like(row)))
            err = probs.max(dim=1)[1].ne(y).double().sum()
            metrics = torch.stack([row.sum(), err, bad.double().sum()])
            if metrics_out is not None:
                metrics = metrics_out.copy_(metrics)
        return probs, None, metrics

    # ---- trunk, staged exactly like the C entries ----
    def fc_buffers(self, model, B, depth, train):
        return CpuFcBuffers(model, B, depth, train)

    def fc_input_layer(self, buf):
        p = buf.model.stacked
        buf.H[0] = F.linear(buf.x, p["w_in"], p["b_in"])

    def fc_block_linear(self, buf, i):
        p = buf.model.stacked
        z = torch.relu(F.linear(buf.H[i], p["w"][i], p["b"][i]))
        buf.Z[i if buf.train else 0] = z
        if buf.train:
            buf.bn_sums[0] = z.double().sum(0)
            buf.bn_sums[1] = (z.double() ** 2).sum(0)

    def fc_block_norm(self, buf, i, global_batch):
        m, p = buf.model, buf.model.stacked
        z = buf.Z[i if buf.train else 0]
        if buf.train:
            m.num_batches_tracked[i] += 1  # (the kernels that update the running statistics do this: dun_fc_ctx.num_batches_tracked)
            mean = buf.bn_sums[0] / global_batch
            var = (buf.bn_sums[1] / global_batch - mean ** 2).clamp(min=0)
            buf.bn_mean[i] = mean.float()
            buf.bn_rstd[i] = (1.0 / torch.sqrt(var + m.bn_eps)).float()
            mom = m.bn_momentum
            m.running_mean[i] = ((1 - mom) * m.running_mean[i].double() + mom * mean).float()
            m.running_var[i] = ((1 - mom) * m.running_var[i].double()
                                + mom * var * global_batch / (global_batch - 1.0)).float()
            mean_f, rstd_f = buf.bn_mean[i], buf.bn_rstd[i]
        else:
            mean_f, rstd_f = m.running_mean[i], 1.0 / torch.sqrt(m.running_var[i] + m.bn_eps)
        blk = (z - mean_f) * (p["gamma"][i] * rstd_f) + p["beta"][i]
        buf.H[i + 1] = buf.H[i] + blk if m.residual else blk
        if buf.head_fused:  # the kernels' row-owned pass writes act[i+1] (and the likelihood terms) here
            self.fc_head_depth(buf, i + 1)

    def _likelihood_terms(self, buf, d):
        """What the head kernels add when the context carries an objective (dun_fc_ctx.obj_*)."""
        o = buf._obj
        if o is None:
            return
        r = o.y.reshape(-1) - buf.act[d].reshape(-1)
        o.partials[d, 0] = (r.double() ** 2).sum()
        if o.w is not None:
            wt = F.softmax(o.w.float(), dim=0) if o.w_is_logits else o.w.float()
            sig = o.log_std.exp().clamp(min=1e-4)
            o.dact[d] = (-(wt[d] * o.inv_batch) * r / sig ** 2).reshape(o.dact[d].shape)

    def fc_head(self, buf):
        p = buf.model.stacked
        buf.act.copy_(F.linear(buf.H, p["w_out"], p["b_out"]))
        for d in range(buf.depth + 1):
            self._likelihood_terms(buf, d)

    def fc_head_depth(self, buf, d):
        p = buf.model.stacked
        buf.act[d].copy_(F.linear(buf.H[d], p["w_out"], p["b_out"]))
        self._likelihood_terms(buf, d)

    def fc_forward(self, buf):
        self.fc_input_layer(buf)
        if buf.head_fused:
            self.fc_head_depth(buf, 0)
        for i in range(buf.depth):
            self.fc_block_linear(buf, i)
            self.fc_block_norm(buf, i, buf.B)
        if not buf.head_fused:
            self.fc_head(buf)

    def fc_bwd_begin(self, buf):
        m, g = buf.model, buf.model.stacked_grad
        D, B, O = buf.dact.shape
        g["w_out"].copy_(buf.dact.reshape(D * B, O).t() @ buf.H.reshape(D * B, -1))
        g["b_out"].copy_(buf.dact.reshape(D * B, O).sum(0))
        buf.G.copy_(buf.dact[buf.depth] @ m.stacked["w_out"])

    def fc_bwd_block_reduce(self, buf, i):
        g = buf.model.stacked_grad
        xh = (buf.Z[i] - buf.bn_mean[i]) * buf.bn_rstd[i]
        buf.bn_sums[0] = buf.G.double().sum(0)
        buf.bn_sums[1] = (buf.G * xh).double().sum(0)
        g["beta"][i] = buf.bn_sums[0].float()
        g["gamma"][i] = buf.bn_sums[1].float()

    def fc_bwd_block_dx(self, buf, i, global_batch):
        m, p, g = buf.model, buf.model.stacked, buf.model.stacked_grad
        z = buf.Z[i]
        xh = (z - buf.bn_mean[i]) * buf.bn_rstd[i]
        mg, mgx = (buf.bn_sums[0] / global_batch).float(), (buf.bn_sums[1] / global_batch).float()
        dU = (z > 0).float() * (p["gamma"][i] * buf.bn_rstd[i]) * (buf.G - mg - xh * mgx)
        buf.dU.copy_(dU)
        g["b"][i] = dU.sum(0)
        new = dU @ p["w"][i] + buf.dact[i] @ p["w_out"]
        buf.G.copy_(buf.G + new if m.residual else new)

    def fc_bwd_block_dw(self, buf, i):
        buf.model.stacked_grad["w"][i] = buf.dU.t() @ buf.H[i]

    def fc_bwd_block_apply(self, buf, i, global_batch):
        self.fc_bwd_block_dx(buf, i, global_batch)
        self.fc_bwd_block_dw(buf, i)

    def fc_bwd_end(self, buf):
        g = buf.model.stacked_grad
        g["w_in"].copy_(buf.G.t() @ buf.x)
        g["b_in"].copy_(buf.G.sum(0))

    def fc_backward(self, buf):
        self.fc_bwd_begin(buf)
        for i in range(buf.model.n_layers - 1, -1, -1):
            self.fc_bwd_block_reduce(buf, i)
            self.fc_bwd_block_apply(buf, i, buf.B)
        self.fc_bwd_end(buf)

    # ---- head / optimiser ----
    def head_forward(self, H, w, b):
        return F.linear(H, w, b)

    def head_backward(self, H, w, dact, need_dH=True):
        return dact.t() @ H, dact.sum(0), (dact @ w if need_dH else None)

    def label_gate(self, y, C):
        bad = ((y < 0) | (y >= C)) & (y != -100)
        return bad.sum().to(torch.float64).reshape(1)

    def sgd_step(self, p, g, buf, lr, momentum, weight_decay):
        ge = g + weight_decay * p
        buf.mul_(momentum).add_(ge)
        p.sub_(lr * buf)
