"""Tensor-level entry to the HIP kernels (through the C ABI in include/dun_hip.h).

Everything the rest of the package computes on the hot path goes through the object returned by
``backend()``.  The product backend is ``HipBackend``; it has no CPU path and raises on CPU tensors.
``set_backend`` exists so that the multi-process CPU tests can drive the host-side sharding logic with a
stand-in (tests/cpu_backend.py); nothing in the package itself ever installs another backend.
"""
import ctypes as C

import torch

from . import _lib as L


class FcBuffers(object):
    """Device buffers of one residual-MLP trunk at a fixed (batch, depth, mode); owns the C context struct."""

    def __init__(self, model, B, depth, train):
        dev = model.flat_params.device
        W, O, din, Lyr = model.width, model.output_dim, model.input_dim, model.n_layers
        self.B, self.depth, self.train = B, depth, train
        f = dict(device=dev, dtype=torch.float32)
        self.H = torch.empty(depth + 1, B, W, **f)
        self.Z = torch.empty((depth if train else min(depth, 1)), B, W, **f)
        self.act = None  # a fresh [depth+1,B,O] tensor per forward (new_act): callers may keep what they get
        self._act_shape = (depth + 1, B, O)
        self.bn_mean = torch.empty(max(Lyr, 1), W, **f)
        self.bn_rstd = torch.empty(max(Lyr, 1), W, **f)
        self.bn_sums = torch.zeros(2, W, device=dev, dtype=torch.float64)
        n_ws = L.lib().dun_fc_workspace_floats(B, din, W, O, Lyr)
        self.ws = torch.empty(n_ws, **f)
        self.G = None
        self.dU = None
        self.x = None
        self.dact = None
        self._obj = None
        self.generation = 0  # bumped by every forward that writes these buffers
        self._dwpart = self._dbpart = None
        c = L.FcCtx()
        c.B, c.din, c.W, c.O, c.L = B, din, W, O, Lyr
        c.residual, c.train, c.depth = int(model.residual), int(train), depth
        c.bn_eps, c.bn_momentum = model.bn_eps, model.bn_momentum
        for k in ("w_in", "b_in", "w_out", "b_out", "w", "b", "gamma", "beta"):
            setattr(c.p, k, L.f32(model.stacked[k]))
            setattr(c.g, k, L.f32(model.stacked_grad[k]))
        c.running_mean, c.running_var = L.f32(model.running_mean), L.f32(model.running_var)
        c.H, c.Z = L.f32(self.H), (L.f32(self.Z) if self.Z.numel() else None)
        c.bn_mean, c.bn_rstd, c.bn_sums = L.f32(self.bn_mean), L.f32(self.bn_rstd), L.ptr(self.bn_sums)
        c.ws, c.ws_floats = L.f32(self.ws), n_ws
        # nn.BatchNorm1d.num_batches_tracked is advanced by the kernels that update the running statistics
        nbt = getattr(model, "num_batches_tracked", None)
        c.num_batches_tracked = nbt.data_ptr() if (nbt is not None and nbt.numel() and nbt.is_cuda) else None
        # hand-off words of the one-launch small-net kernels: zeroed once here, owned by the kernels afterwards
        n_sync = L.lib().dun_fc_sync_bytes(B, W)
        self.sync = torch.zeros(n_sync // 4, device=dev, dtype=torch.int32) if n_sync else None
        if self.sync is not None:
            c.sync, c.sync_bytes = L.ptr(self.sync), n_sync
        self.c = c

    def new_act(self):
        self.act = torch.empty(self._act_shape, device=self.H.device, dtype=torch.float32)
        self.c.act = L.f32(self.act)

    def set_input(self, x):
        self.x = x
        self.c.x = L.f32(x)

    def ensure_backward(self):
        if self.G is None:
            dev = self.H.device
            self.G = torch.empty(self.B, self.H.shape[2], device=dev, dtype=torch.float32)
            self.dU = torch.empty_like(self.G)
            self.c.G, self.c.dU = L.f32(self.G), L.f32(self.dU)

    def set_dact(self, dact):
        self.dact = dact
        self.c.dact = L.f32(dact)
        # Backward with the very d loss / d act the forward's head kernels wrote: the head's dw / db partial sums of
        # that forward are valid, backward adds them up instead of re-reading every hidden state
        o = self._obj
        self.c.head_grads_from_forward = int(o is not None and o.dact is not None and self.c.obj_dwpart is not None
                                             and dact.data_ptr() == o.dact.data_ptr() and dact.numel() == o.dact.numel())

    def set_objective(self, spec):
        """Fold the Gaussian likelihood pass into the head kernels of the next forward (spec: probability.HeadObjective,
        or None for a plain forward).  Its outputs are fresh tensors per forward, handed over through the spec."""
        c = self.c
        c.obj_dwpart = c.obj_dbpart = None
        c.head_grads_from_forward = 0
        c.obj_fin_result = c.obj_fin_sums = c.obj_fin_q_logits = c.obj_fin_logprior = c.obj_fin_prior_ll = None
        c.obj_fin_defer = 0
        if spec is None:
            c.obj_y = c.obj_w = c.obj_log_std = c.obj_dact = c.obj_partials = None
            self._obj = None
            return
        D, B, O = self._act_shape
        nparts = L.lib().dun_fc_obj_num_partials(B, O)
        if nparts <= 0:
            raise RuntimeError("the likelihood pass inside the head needs one output dimension")
        dev = self.H.device
        spec.partials = torch.empty(D, nparts, device=dev, dtype=torch.float64)
        spec.dact = torch.empty(self._act_shape, device=dev, dtype=torch.float32) if spec.w is not None else None
        c.obj_y, c.obj_log_std = L.f32(spec.y), L.f32(spec.log_std)
        c.obj_w = None if spec.w is None else L.f32(spec.w)
        c.obj_w_is_logits, c.obj_inv_batch = int(spec.w_is_logits), float(spec.inv_batch)
        c.obj_dact = None if spec.dact is None else L.f32(spec.dact)
        c.obj_partials = L.ptr(spec.partials)
        self._obj = spec  # keeps the tensors behind the raw pointers alive until the next forward
        fin = getattr(spec, "fin", None)
        spec.res = spec.sums = None
        if fin is not None:
            # where the forward is ONE launch (small nets) the workgroup that delivers the last partial sums also runs
            # the depth-level finalisation: the step needs no dun_depth_finalize_partials launch
            spec.res = torch.empty(L.res_floats(D), device=dev, dtype=torch.float32)
            spec.sums = torch.empty(D, device=dev, dtype=torch.float64)
            c.obj_fin_result, c.obj_fin_sums = L.f32(spec.res), L.ptr(spec.sums)
            c.obj_fin_q_logits = None if fin["q_logits"] is None else L.f32(fin["q_logits"])
            c.obj_fin_logprior = L.f32(fin["logprior"])
            c.obj_fin_prior_ll = None
            c.obj_fin_N_train, c.obj_fin_batch = float(fin["N_train"]), float(fin["batch"])
            c.obj_fin_beta, c.obj_fin_eps, c.obj_fin_objective = float(fin["beta"]), float(fin["eps"]), int(fin["objective"])
            # deferred: the backward of this context, which the caller runs next, finishes the objective in a spare
            # workgroup of its weight-gradient launch (spec.res is valid once that backward has been enqueued)
            c.obj_fin_defer = int(bool(fin.get("defer")) and self.train)
            if not L.lib().dun_fc_objective_in_forward(C.byref(c)):
                c.obj_fin_result = c.obj_fin_sums = c.obj_fin_q_logits = c.obj_fin_logprior = None
                spec.res = spec.sums = None
        if spec.dact is not None and self.train and L.lib().dun_fc_head_grads_in_forward(C.byref(c)):
            # the head's weight / bias gradient is accumulated by the forward's head kernels (per-workgroup partials,
            # kept with the buffers: [D, nparts, W] floats, 214 MB at the wide configuration)
            if self._dwpart is None:
                self._dwpart = torch.empty(D, nparts, self.H.shape[2], device=dev, dtype=torch.float32)
                self._dbpart = torch.empty(D, nparts, device=dev, dtype=torch.float32)
            c.obj_dwpart, c.obj_dbpart = L.f32(self._dwpart), L.f32(self._dbpart)

    @property
    def head_fused(self):
        """block_norm(i) also writes act[i+1] (dun_fc_head_is_fused): a staged caller runs fc_head_depth for depth 0 only."""
        return bool(L.lib().dun_fc_head_is_fused(C.byref(self.c)))


class HipBackend(object):
    name = "hip"

    # ---- probability module -------------------------------------------------------------------------
    def ll_pass(self, act, y, regression, log_std, w, w_is_logits, inv_batch, want_dact=True, want_ll=False,
                reduce=True):
        """One streaming pass: per-depth sums (double [D]) [+ dact] [+ ll[D,B]].  With ``reduce=False`` the first result
        is the per-workgroup partials (double [D, nparts]) for ``depth_finalize(..., partials=True)``, which adds them
        up inside the finalisation launch (single-device jobs: nothing to exchange in between)."""
        lib = L.lib()
        D, B = act.shape[0], act.shape[1]
        dev = act.device
        if y.numel() != B:
            raise RuntimeError("targets (%d) do not match the batch of act_vec %s" % (y.numel(), tuple(act.shape)))
        if w is not None and w.numel() != D:
            raise RuntimeError("depth weights (%d) do not match the number of depths of act_vec (%d)" % (w.numel(), D))
        if regression and (log_std is None or log_std.numel() != 1):
            raise RuntimeError("the Gaussian likelihood pass needs a log_std of one element")
        dact = torch.empty_like(act) if want_dact else None
        ll = torch.empty(D, B, device=dev, dtype=torch.float32) if want_ll else None
        s = L.stream_ptr(dev)
        if regression:
            if act.dim() == 3 and act.shape[2] != 1:
                # same failure the reference hits at src/probability.py:171 (.view(depth, batch_size))
                raise RuntimeError("shape '[%d, %d]' is invalid for input of size %d" % (D, B, act.numel()))
            nparts = lib.dun_ll_num_partials(B, 1)
            partials = torch.empty(D, nparts, device=dev, dtype=torch.float64)
            L.check(lib.dun_ll_gauss(L.f32(act), L.f32(y), L.f32(log_std), None if w is None else L.f32(w),
                                     int(w_is_logits), D, B, float(inv_batch), None if dact is None else L.f32(dact),
                                     None if ll is None else L.f32(ll), L.ptr(partials), s), "dun_ll_gauss")
        else:
            Cn = act.shape[2]
            if y.dtype != torch.int64:
                raise RuntimeError("classification targets must be int64")
            nparts = lib.dun_ll_num_partials(B, max(Cn, 2))
            partials = torch.empty(D, nparts, device=dev, dtype=torch.float64)
            L.check(lib.dun_ll_softmax(L.f32(act), L.ptr(y), None if w is None else L.f32(w), int(w_is_logits), D, B,
                                       Cn, float(inv_batch), None if dact is None else L.f32(dact),
                                       None if ll is None else L.f32(ll), L.ptr(partials), s), "dun_ll_softmax")
        if not reduce:
            return partials, dact, ll
        if nparts == 1:
            sums = partials.view(D)  # one workgroup per depth: its partial is the sum, no reduction launch
        else:
            sums = torch.empty(D, device=dev, dtype=torch.float64)
            L.check(lib.dun_ll_reduce(L.ptr(partials), D, nparts, L.ptr(sums), s), "dun_ll_reduce")
        return sums, dact, ll

    def ll_reduce(self, partials):
        """sums[d] = sum_j partials[d, j] (what a batch-sharded job all-reduces before depth_finalize)."""
        D, nparts = partials.shape
        if nparts == 1:
            return partials.view(D)
        sums = torch.empty(D, device=partials.device, dtype=torch.float64)
        L.check(L.lib().dun_ll_reduce(L.ptr(partials), D, nparts, L.ptr(sums), L.stream_ptr(partials.device)),
                "dun_ll_reduce")
        return sums

    def depth_finalize(self, sums, gaussian, q_logits, logprior, prior_ll, log_std, N_train, batch, beta, eps,
                       objective, partials=False):
        """-> result block, or (result block, sums [D]) when `sums` holds unreduced partials [D, nparts]."""
        D = sums.shape[0]
        res = torch.empty(L.res_floats(D), device=sums.device, dtype=torch.float32)
        if partials:
            nparts = sums.shape[1]
            out_sums = torch.empty(D, device=sums.device, dtype=torch.float64)
            L.check(L.lib().dun_depth_finalize_partials(
                L.ptr(sums), nparts, L.ptr(out_sums), int(gaussian), None if q_logits is None else L.f32(q_logits),
                L.f32(logprior), None if prior_ll is None else L.f32(prior_ll),
                None if log_std is None else L.f32(log_std), D, float(N_train), float(batch), float(beta), float(eps),
                int(objective), L.f32(res), L.stream_ptr(sums.device)), "dun_depth_finalize_partials")
            return res, out_sums
        L.check(L.lib().dun_depth_finalize(
            L.ptr(sums), int(gaussian), None if q_logits is None else L.f32(q_logits), L.f32(logprior),
            None if prior_ll is None else L.f32(prior_ll), None if log_std is None else L.f32(log_std), D,
            float(N_train), float(batch), float(beta), float(eps), int(objective), L.f32(res),
            L.stream_ptr(sums.device)), "dun_depth_finalize")
        return res

    def predict(self, act, q, regression, get_std=False, y=None, log_std=None, metrics_out=None):
        """Depth-marginal prediction; with y also the metric sums (double [2], written into `metrics_out` if given).
        Regression with ``log_std`` but without ``y``: the returned std is the total predictive one,
        sqrt(model_std^2 + exp(log_std)^2), from the same launch."""
        lib = L.lib()
        dev = act.device
        D = act.shape[0]
        s = L.stream_ptr(dev)
        q = q.detach().to(torch.float32).contiguous()
        if q.numel() != D:
            # the reference's (q * preds) broadcast raises here too (e.g. predict(x, depth=k) with a full-length posterior)
            raise RuntimeError("The size of the depth posterior (%d) must match the number of depths of act_vec (%d)"
                               % (q.numel(), D))
        if y is not None and y.numel() != (act[0].numel() if regression else act.shape[1]):
            raise RuntimeError("targets (%d) do not match the batch of act_vec %s" % (y.numel(), tuple(act.shape)))
        metrics = None
        if act.shape[1] == 0 and y is None:  # empty inference batch: empty outputs, like the reference's torch ops
            empty = torch.empty(act.shape[1:], device=dev, dtype=torch.float32)
            return empty, (torch.empty_like(empty) if (regression and get_std) else None), None
        if regression:
            n = act[0].numel()
            mean = torch.empty(act.shape[1:], device=dev, dtype=torch.float32)
            std = torch.empty_like(mean) if (get_std or y is not None) else None
            partials = None
            if y is not None:
                nparts = lib.dun_predict_num_partials(n, 1)
                partials = torch.empty(2, nparts, device=dev, dtype=torch.float64)
                metrics = metrics_out if metrics_out is not None else torch.empty(2, device=dev, dtype=torch.float64)
            L.check(lib.dun_predict_gauss(L.f32(act), L.f32(q), D, n, L.f32(mean), None if std is None else L.f32(std),
                                          None if y is None else L.f32(y), None if log_std is None else L.f32(log_std),
                                          None if partials is None else L.ptr(partials),
                                          None if metrics is None else L.ptr(metrics), s), "dun_predict_gauss")
            return mean, std, metrics
        B, Cn = act.shape[1], act.shape[2]
        probs = torch.empty(B, Cn, device=dev, dtype=torch.float32)
        partials = None
        if y is not None:
            if y.dtype != torch.int64:
                raise RuntimeError("classification targets must be int64")
            nparts = lib.dun_predict_num_partials(B, max(Cn, 2))
            partials = torch.empty(3, nparts, device=dev, dtype=torch.float64)
            # [nll sum, error count, count of labels outside [0,C) other than -100]
            metrics = metrics_out if metrics_out is not None else torch.empty(3, device=dev, dtype=torch.float64)
            if metrics.numel() != 3:
                raise RuntimeError("softmax metrics need a double[3] buffer")
        L.check(lib.dun_predict_softmax(L.f32(act), L.f32(q), D, B, Cn, L.f32(probs), None if y is None else L.ptr(y),
                                        None if partials is None else L.ptr(partials),
                                        None if metrics is None else L.ptr(metrics), s), "dun_predict_softmax")
        return probs, None, metrics

    def predict_step(self, act, q_logits, y, log_std, objective, out):
        """End of a regression training step in ONE launch: depth-marginal mean / std of `act` under q = softmax(q_logits)
        (the post-step q(d)), the metric sums, q itself and the step's objective.  `out`: double [3] (device, or pinned
        host memory) receiving [objective, nll_sum, squared-error sum].  -> q (float [D]).  Falls back to None for batches that need the
        two-stage metric reduction."""
        lib = L.lib()
        D, n = act.shape[0], act[0].numel()
        if lib.dun_predict_num_partials(n, 1) != 1 or q_logits.numel() != D:
            return None
        dev = act.device
        mean = torch.empty(n, device=dev, dtype=torch.float32)
        std = torch.empty(n, device=dev, dtype=torch.float32)
        q = torch.empty(D, device=dev, dtype=torch.float32)
        m = out[1:3]
        if out.numel() > 3:  # out[3]: completion word (pinned host memory a captured step polls instead of synchronising)
            L.check(lib.dun_predict_gauss_step_signal(L.f32(act), L.f32(q_logits), 1, D, n, L.f32(mean), L.f32(std), L.f32(y),
                                                      L.f32(log_std), L.out_ptr(m), L.f32(q), L.f32(objective),
                                                      L.out_ptr(out), L.out_ptr(out[3:4]), L.stream_ptr(dev)),
                    "dun_predict_gauss_step_signal")
            return q
        L.check(lib.dun_predict_gauss_step(L.f32(act), L.f32(q_logits), 1, D, n, L.f32(mean), L.f32(std), L.f32(y),
                                           L.f32(log_std), L.out_ptr(m), L.out_ptr(m), L.f32(q), L.f32(objective),
                                           L.out_ptr(out), L.stream_ptr(dev)), "dun_predict_gauss_step")
        return q

    # ---- trunk ----------------------------------------------------------------------------------------
    def fc_buffers(self, model, B, depth, train):
        return FcBuffers(model, B, depth, train)

    def _fc(self, name, buf, *args):
        fn = getattr(L.lib(), name)
        L.check(fn(C.byref(buf.c), *args, L.stream_ptr(buf.H.device)), name)

    def fc_forward(self, buf):
        self._fc("dun_fc_forward", buf)

    def fc_backward(self, buf):
        self._fc("dun_fc_backward", buf)

    def fc_input_layer(self, buf):
        self._fc("dun_fc_input_layer", buf)

    def fc_block_linear(self, buf, i):
        self._fc("dun_fc_block_linear", buf, i)

    def fc_block_norm(self, buf, i, global_batch):
        self._fc("dun_fc_block_norm", buf, i, float(global_batch))

    def fc_block_apply(self, buf, i):
        self._fc("dun_fc_block_apply", buf, i)

    def fc_head(self, buf):
        self._fc("dun_fc_head", buf)

    def fc_head_depth(self, buf, d):
        self._fc("dun_fc_head_depth", buf, d)

    def fc_bwd_begin(self, buf):
        self._fc("dun_fc_bwd_begin", buf)

    def fc_bwd_block_reduce(self, buf, i):
        self._fc("dun_fc_bwd_block_reduce", buf, i)

    def fc_bwd_block_apply(self, buf, i, global_batch):
        self._fc("dun_fc_bwd_block_apply", buf, i, float(global_batch))

    def fc_bwd_block_dx(self, buf, i, global_batch):
        self._fc("dun_fc_bwd_block_dx", buf, i, float(global_batch))

    def fc_bwd_block_dw(self, buf, i):
        self._fc("dun_fc_bwd_block_dw", buf, i)

    def fc_bwd_end(self, buf):
        self._fc("dun_fc_bwd_end", buf)

    # ---- shared head over all exits -------------------------------------------------------------------
    def head_forward(self, H, w, b):
        rows, K = H.shape[0], H.shape[1]
        O = w.shape[0]
        act = torch.empty(rows, O, device=H.device, dtype=torch.float32)
        L.check(L.lib().dun_head_forward(L.f32(H), L.f32(w), L.f32(b), rows, K, O, L.f32(act), L.stream_ptr(H.device)),
                "dun_head_forward")
        return act

    def head_backward(self, H, w, dact, need_dH=True):
        lib = L.lib()
        rows, K = H.shape[0], H.shape[1]
        O = w.shape[0]
        dw = torch.empty_like(w)
        db = torch.empty(O, device=H.device, dtype=torch.float32)
        dH = torch.empty_like(H) if need_dH else None
        n_ws = lib.dun_head_backward_workspace_floats(rows, K, O)
        ws = torch.empty(n_ws, device=H.device, dtype=torch.float32)
        L.check(lib.dun_head_backward(L.f32(H), L.f32(w), L.f32(dact), rows, K, O, L.f32(dw), L.f32(db),
                                      None if dH is None else L.f32(dH), L.f32(ws), n_ws, L.stream_ptr(H.device)),
                "dun_head_backward")
        return dw, db, dH

    # ---- optimiser ------------------------------------------------------------------------------------
    def sgd_step(self, p, g, buf, lr, momentum, weight_decay):
        L.check(L.lib().dun_sgd_step(L.f32(p), L.f32(g), L.f32(buf), p.numel(), float(lr), float(momentum),
                                     float(weight_decay), L.stream_ptr(p.device)), "dun_sgd_step")

    def copy_pair(self, dst0, src0, dst1, src1):
        """dst0 <- src0 and dst1 <- src1 (same shapes / dtypes, contiguous device tensors) in ONE launch."""
        L.check(L.lib().dun_copy_pair(L.ptr(dst0), L.ptr(src0), dst0.numel() * dst0.element_size(), L.ptr(dst1),
                                      L.ptr(src1), dst1.numel() * dst1.element_size(), L.stream_ptr(dst0.device)),
                "dun_copy_pair")

    def copy_pair_indirect(self, dst0, dst1, table):
        """dst0 <- *table[0], dst1 <- *table[1]: the source addresses (int64 [2], pinned host or device memory) are read
        when the launch runs, so a captured step keeps its input copies inside the graph."""
        L.check(L.lib().dun_copy_pair_indirect(L.ptr(dst0), dst0.numel() * dst0.element_size(), L.ptr(dst1),
                                               dst1.numel() * dst1.element_size(), L.out_ptr(table),
                                               L.stream_ptr(dst0.device)), "dun_copy_pair_indirect")

    def sgd_step_multi(self, triples, lr, momentum, weight_decay, gate=None):
        """The same update over several (param, grad, momentum buffer) triples, DUN_SGD_MAX_SEGMENTS per launch.
        ``gate``: a device double (``label_gate``); when it is positive the launches change nothing."""
        lib = L.lib()
        for k in range(0, len(triples), L.SGD_MAX_SEGMENTS):
            chunk = triples[k:k + L.SGD_MAX_SEGMENTS]
            segs = (L.SgdSegment * len(chunk))()
            for sg, (p, g, b) in zip(segs, chunk):
                sg.p, sg.g, sg.buf, sg.n = L.f32(p), L.f32(g), L.f32(b), p.numel()
            if gate is None:
                L.check(lib.dun_sgd_step_multi(segs, len(chunk), float(lr), float(momentum), float(weight_decay),
                                               L.stream_ptr(chunk[0][0].device)), "dun_sgd_step_multi")
            else:
                L.check(lib.dun_sgd_step_multi_gated(segs, len(chunk), float(lr), float(momentum), float(weight_decay),
                                                     L.ptr(gate), L.stream_ptr(chunk[0][0].device)),
                        "dun_sgd_step_multi_gated")

    def label_gate(self, y, C):
        """Device double [1]: how many labels are outside [0, C) and not the ignored -100 (nn.CrossEntropyLoss raises on
        those before any update; the step's SGD launches take this count as their gate)."""
        y = y.contiguous()
        if y.dtype != torch.int64:
            raise TypeError("label_gate: int64 labels, got %s" % y.dtype)
        out = torch.empty(1, device=y.device, dtype=torch.float64)
        L.check(L.lib().dun_label_count_bad(L.ptr(y), y.numel(), int(C), L.ptr(out), L.stream_ptr(y.device)),
                "dun_label_count_bad")
        return out

    # ---- raw GEMM (tests / benchmarks) ----------------------------------------------------------------
    def gemm(self, A, B, M, N, K, a_kcontig=True, b_kcontig=True, bias=None, relu=False, R=None, U=None, V=None,
             colstats=False, out=None, splitk_ws=None, bn=None):
        """bn = (Z [M,N], mean [N], rstd [N]): also return the per-row-tile column sums of C and C*(Z-mean)*rstd
        (float [2][T][N], the BatchNorm backward statistics the trunk's dX GEMM emits for the block below)."""
        dev = A.device
        Cm = out if out is not None else torch.empty(M, N, device=dev, dtype=torch.float32)
        d = L.GemmDesc()
        d.A, d.B, d.C = L.f32(A), L.f32(B), L.f32(Cm)
        d.M, d.N, d.K = M, N, K
        d.lda, d.ldb, d.ldc = A.shape[1], B.shape[1], N
        d.a_kcontig, d.b_kcontig = int(a_kcontig), int(b_kcontig)
        d.bias = None if bias is None else L.f32(bias)
        d.relu = int(relu)
        if R is not None:
            d.R, d.ldr = L.f32(R), R.shape[1]
        if U is not None:
            d.U, d.V, d.O, d.ldv = L.f32(U), L.f32(V), U.shape[1], V.shape[1]
        stats = None
        if colstats:
            stats = torch.empty(2, L.lib().dun_gemm_colstat_tiles(M, N), N, device=dev, dtype=torch.float64)
            d.colstats = L.ptr(stats)
        if splitk_ws is not None:
            d.splitk_ws, d.splitk_ws_floats = L.f32(splitk_ws), splitk_ws.numel()
        bnstats = None
        if bn is not None:
            bnstats = torch.empty(2, L.lib().dun_gemm_colstat_tiles(M, N), N, device=dev, dtype=torch.float32)
            d.bnz, d.bn_mean, d.bn_rstd, d.bnstats, d.ldz = L.f32(bn[0]), L.f32(bn[1]), L.f32(bn[2]), L.f32(bnstats), N
        L.check(L.lib().dun_gemm_f32(C.byref(d), L.stream_ptr(dev)), "dun_gemm_f32")
        if bn is not None:
            return Cm, bnstats
        return (Cm, stats) if colstats else Cm


_backend = None


def backend():
    global _backend
    if _backend is None:
        L.lib()  # fail loudly, now, if the HIP library was never built
        _backend = HipBackend()
    return _backend


def set_backend(b):
    """Test seam only (see module docstring)."""
    global _backend
    _backend = b
