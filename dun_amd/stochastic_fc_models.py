"""Residual-MLP / plain-MLP DUN trunks with the reference's constructor and attribute surface.

Reference: src/DUN/stochastic_fc_models.py:7-80.  ``forward(x, depth=None)`` returns ``act_vec [D,B,O]``
(the north-star's "forward_get_acts"; an alias of that name is provided).  All arithmetic of ``forward``
and its backward runs in the HIP trunk kernels (dun_amd/csrc/trunk.hip, gemm.hip, head.hip): the module's
parameters are views into one flat fp32 buffer that the kernels, the fused SGD step and the gradient
all-reduce address directly; gradients land in a second flat buffer whose views are the parameters' ``.grad``.
"""
import torch
import torch.nn as nn

from . import kernels
from .layers import bern_MLPBlock, bern_MLPBlock_nores

_SEG_ALIGN = 64  # floats: every segment of the flat buffers starts on a 256-byte boundary
_KINDS = ("w_in", "b_in", "w_out", "b_out", "w", "b", "gamma", "beta")


def _round_up(n, a):
    return (n + a - 1) // a * a


class _FcForward(torch.autograd.Function):
    """Autograd node of the whole trunk.  Parameter gradients are written by the kernels straight into the
    flat gradient buffer (``p.grad`` views), not returned to autograd, and overwrite what was there."""

    @staticmethod
    def forward(ctx, model, x, depth, objective, *params):
        buf = model._run_forward(x, depth, True, objective)
        ctx.model, ctx.buf, ctx.generation = model, buf, buf.generation
        # The returned tensor becomes this node's output (grad_fn -> node -> parameters -> model).  Left in the buffer
        # object it would close model -> buffers -> act -> node -> model through the C++ autograd graph, which
        # Python's collector cannot see: a dropped model kept its activation buffers (55 GB at C3) allocated.
        act, buf.act = buf.act, None
        return act

    @staticmethod
    def backward(ctx, dact):
        model, buf = ctx.model, ctx.buf
        if buf is None:
            raise RuntimeError("the trunk's backward ran already (its buffers are reused by the next forward)")
        ctx.model = ctx.buf = None  # a kept act_vec does not pin the buffers
        if buf.generation != ctx.generation:
            # H / Z / BatchNorm statistics live in ONE buffer set per (batch, depth, mode): a later train-mode forward
            # of the same batch size (also under no_grad, e.g. a sweep with train_bn=True) has overwritten what this
            # backward needs.  Gradients are also written, not accumulated (no micro-batch accumulation).
            raise RuntimeError("the trunk's activation buffers were reused by a later train-mode forward of the same batch "
                               "size before this backward ran; run backward before the next forward")
        model._run_backward(buf, dact)
        return (None, None, None, None) + (None,) * len(model._param_list)


class arq_uncert_fc_resnet(nn.Module):
    residual = True
    _block_cls = bern_MLPBlock

    def __init__(self, input_dim, output_dim, width, n_layers, w_prior=None, BMA_prior=False):
        super(arq_uncert_fc_resnet, self).__init__()
        self.input_dim = input_dim
        self.output_dim = output_dim
        # same construction order as the reference, so an identical torch seed gives identical initial weights
        self.input_layer = nn.Linear(self.input_dim, width)
        self.output_layer = nn.Linear(width, self.output_dim)
        self.n_layers = n_layers
        self.width = width
        self.w_prior = w_prior
        self.BMA_prior = BMA_prior
        if not isinstance(self.w_prior, list) and self.w_prior is not None:
            self.w_prior = [self.w_prior for _ in range(self.n_layers + 1)]
        self.stochstic_layers = nn.ModuleList()
        for _ in range(self.n_layers):
            self.stochstic_layers.append(self._block_cls(width))
        self.bn_eps, self.bn_momentum = 1e-5, 0.1
        self._dist = None      # set by dun_amd.distributed.shard(): batch-sharded data parallelism
        self._fc_cache = {}
        self._flatten()

    # ---- flat parameter / gradient storage ---------------------------------------------------------
    def _segments(self):
        W, L, O, din = self.width, self.n_layers, self.output_dim, self.input_dim
        sizes = dict(w_in=W * din, b_in=W, w_out=O * W, b_out=O, w=L * W * W, b=L * W, gamma=L * W, beta=L * W)
        shapes = dict(w_in=(W, din), b_in=(W,), w_out=(O, W), b_out=(O,), w=(L, W, W), b=(L, W), gamma=(L, W),
                      beta=(L, W))
        offs, o = {}, 0
        for k in _KINDS:
            offs[k] = o
            o += _round_up(sizes[k], _SEG_ALIGN)
        return sizes, shapes, offs, max(o, _SEG_ALIGN)

    def _module_params(self):
        """(kind, layer index or None, Parameter) in flat-buffer order."""
        out = [("w_in", None, self.input_layer.weight), ("b_in", None, self.input_layer.bias),
               ("w_out", None, self.output_layer.weight), ("b_out", None, self.output_layer.bias)]
        for kind, sub, attr in (("w", 0, "weight"), ("b", 0, "bias"), ("gamma", 2, "weight"), ("beta", 2, "bias")):
            for i, blk in enumerate(self.stochstic_layers):
                out.append((kind, i, getattr(blk.block[sub], attr)))
        return out

    def _flatten(self):
        sizes, shapes, offs, total = self._segments()
        dev = self.input_layer.weight.device
        flat = torch.zeros(total, device=dev, dtype=torch.float32)
        gflat = torch.zeros(total, device=dev, dtype=torch.float32)
        self.stacked, self.stacked_grad = {}, {}
        for k in _KINDS:
            self.stacked[k] = flat[offs[k]:offs[k] + sizes[k]].view(shapes[k])
            self.stacked_grad[k] = gflat[offs[k]:offs[k] + sizes[k]].view(shapes[k])
        self._param_list, self._grad_views = [], []
        with torch.no_grad():
            for kind, i, p in self._module_params():
                dst = self.stacked[kind] if i is None else self.stacked[kind][i]
                gdst = self.stacked_grad[kind] if i is None else self.stacked_grad[kind][i]
                dst.copy_(p.data.to(torch.float32))
                p.data = dst
                p.grad = gdst
                p._dun_flat_owner = self
                self._param_list.append(p)
                self._grad_views.append(gdst)
            L, W = self.n_layers, self.width
            rm = torch.zeros(L, W, device=dev)
            rv = torch.ones(L, W, device=dev)
            nbt = torch.zeros(L, device=dev, dtype=torch.int64)
            for i, blk in enumerate(self.stochstic_layers):
                bn = blk.block[2]
                rm[i].copy_(bn.running_mean)
                rv[i].copy_(bn.running_var)
                nbt[i] = bn.num_batches_tracked
                bn.running_mean = rm[i]
                bn.running_var = rv[i]
                bn.num_batches_tracked = nbt[i]
        self.flat_params, self.flat_grads = flat, gflat
        # captured steps hold the addresses of the buffers they were captured on: a rebuilt flat buffer (load_state_dict,
        # .to()) may land on the old parameter address while its gradient buffer does not, so the wrappers key their
        # graphs on this counter too
        self._flat_generation = self.__dict__.get("_flat_generation", 0) + 1
        self.running_mean, self.running_var, self.num_batches_tracked = rm, rv, nbt
        self._fc_cache = {}
        # every parameter of the module tree lives in the flat buffer (what the whole-step hipGraph relies on)
        self._all_params_flat = all(getattr(q, "_dun_flat_owner", None) is self for q in self.parameters())

    def _apply(self, fn, *args, **kwargs):
        super(arq_uncert_fc_resnet, self)._apply(fn, *args, **kwargs)
        if hasattr(self, "flat_params"):
            self._flatten()  # .cuda()/.to() re-created the tensors: rebuild the flat views on the new device
        return self

    def load_state_dict(self, state_dict, strict=True, **kw):
        out = super(arq_uncert_fc_resnet, self).load_state_dict(state_dict, strict=strict, **kw)
        self._flatten()
        return out

    def __getstate__(self):
        st = self.__dict__.copy()
        st["_fc_cache"] = {}
        st["_dist"] = None
        st.pop("_mode_synced", None)
        for k in ("stacked", "stacked_grad", "_param_list", "_grad_views", "flat_params", "flat_grads",
                  "running_mean", "running_var", "num_batches_tracked"):
            st.pop(k, None)
        return st

    def __setstate__(self, st):
        super(arq_uncert_fc_resnet, self).__setstate__(st)
        self._flatten()

    def train(self, mode=True):
        """nn.Module.train walks the ~30 sub-modules (tens of microseconds per call, and fit / eval / predict call it
        every step like the reference does): skipped when the tree is already in that mode.  The kernels only read
        this module's own flag; the children's flags matter to the torch cold path (vec_forward) and are kept in
        step as long as modes are switched through this module, which is what the wrappers do."""
        if self.training == mode and self.__dict__.get("_mode_synced") == mode:
            return self
        super(arq_uncert_fc_resnet, self).train(mode)
        self.__dict__["_mode_synced"] = mode
        return self

    # ---- weight priors: O(#params) torch reductions, not part of the per-sample path ----
    def apply_prior(self, module, prior_f):
        if isinstance(module, (nn.Linear, nn.Conv2d, nn.Conv1d)):
            self.loglike = self.loglike + prior_f(module.weight)
            self.loglike = self.loglike + prior_f(module.bias)

    def get_w_prior_loglike(self, k=None):
        if k is None:
            k = self.n_layers
        if self.w_prior is None:
            return self.input_layer.bias.data.new_zeros(k + 1)
        self.loglike = 0
        self.input_layer.apply(lambda m: self.apply_prior(m, self.w_prior[0]))
        self.output_layer.apply(lambda m: self.apply_prior(m, self.w_prior[0]))
        self.loglikes = [self.loglike]
        for idx, layer in enumerate(self.stochstic_layers[:k]):
            self.loglike = 0
            layer.apply(lambda m: self.apply_prior(m, self.w_prior[idx + 1]))
            self.loglikes.append(self.loglike)
        loglike_vec = torch.stack(self.loglikes, dim=0)
        if self.BMA_prior:
            loglike_vec = loglike_vec.unsqueeze(0).repeat(k + 1, 1).tril(diagonal=0).sum(dim=1)
        return loglike_vec.sum(dim=0)

    # ---- cold path: explicit 0/1 mask per block, torch ops (plots only) -------------------
    def vec_forward(self, x, vec):
        assert vec.shape[0] == self.n_layers
        x = self.input_layer(x)
        for i in range(self.n_layers):
            x = self.stochstic_layers[i](x, vec[i])
        return self.output_layer(x)

    # ---- hot path ---------------------------------------------------------------------------------------
    def _get_buffers(self, B, depth, train):
        key = (B, depth, bool(train))
        buf = self._fc_cache.get(key)
        if buf is None:
            if len(self._fc_cache) >= 4:  # keep the cache bounded: training + a few inference shapes
                for k in [k for k in self._fc_cache if not k[2]][:1] or list(self._fc_cache)[:1]:
                    del self._fc_cache[k]
            buf = kernels.backend().fc_buffers(self, B, depth, train)
            self._fc_cache[key] = buf
        return buf

    supports_head_objective = True  # forward(x, objective=HeadObjective): likelihood pass inside the head kernels

    def _run_forward(self, x, depth, train, objective=None):
        if x.dim() != 2 or x.shape[1] != self.input_dim:
            raise RuntimeError("expected x of shape [B, %d], got %s" % (self.input_dim, tuple(x.shape)))
        if train and depth > 0:
            # nn.BatchNorm1d in training mode (torch.nn.functional.batch_norm -> _verify_batch_size): a batch of one
            # row has no variance.  The reference's blocks raise this from a ragged last batch of size 1; here the
            # unbiased running-variance update would divide by n - 1 = 0 and poison the running statistics silently.
            dist = self._dist
            n_stat = dist.global_batch(x.shape[0]) if (dist is not None and dist.sync_bn) else x.shape[0]
            if n_stat <= 1:
                raise ValueError("Expected more than 1 value per channel when training, got input size %s"
                                 % (torch.Size([x.shape[0], self.width]),))
        be = kernels.backend()
        x = x.detach().to(torch.float32).contiguous()
        buf = self._get_buffers(x.shape[0], depth, train)
        buf.generation += 1
        buf.new_act()
        buf.set_input(x)
        buf.set_objective(objective if self.output_dim == 1 else None)
        dist = self._dist
        if dist is None or not (train and dist.sync_bn) or depth == 0:
            be.fc_forward(buf)
        else:
            # Staged at the BatchNorm exchanges.  Where block_norm(i) writes act[i+1] itself (row-owned BatchNorm apply
            # + head: buf.head_fused) only depth 0 needs the head kernel; otherwise the shared head runs depth by depth
            # on the hidden state block i started from, with dist.async_forward_bn UNDER the exchange of block i
            gb = dist.global_batch(x.shape[0])
            fused = buf.head_fused
            be.fc_input_layer(buf)
            if fused:
                be.fc_head_depth(buf, 0)
            for i in range(depth):
                be.fc_block_linear(buf, i)
                if dist.async_forward_bn and not fused:
                    h = dist.all_reduce_async(buf.bn_sums)
                    be.fc_head_depth(buf, i)
                    if h is not None:
                        h.wait()
                else:
                    dist.all_reduce(buf.bn_sums)
                    if not fused:
                        be.fc_head_depth(buf, i)
                be.fc_block_norm(buf, i, gb)
            if not fused:
                be.fc_head_depth(buf, depth)
        # (nn.BatchNorm1d.num_batches_tracked of the blocks that ran is advanced by the kernels that update the running
        # statistics: dun_fc_ctx.num_batches_tracked)
        return buf

    def _run_backward(self, buf, dact):
        be = kernels.backend()
        buf.ensure_backward()
        buf.set_dact(dact.detach().to(torch.float32).contiguous())
        dist = self._dist
        if dist is None:
            be.fc_backward(buf)
        else:
            gb = dist.global_batch(buf.B)
            be.fc_bwd_begin(buf)
            # The BatchNorm exchange of block i (16 KB, latency bound) sits between its column reduction and its dU;
            # the weight-gradient GEMM of block i+1 needs neither, so it is deferred to run UNDER that exchange (dU of
            # block i+1 stays in place until bwd_block_dx(i)).  The gradient message of block i+1 follows the small
            # exchange on the single communicator stream and has a dX GEMM + a reduction + a dW GEMM of time before
            # the next small message is queued behind it.
            # Weight gradients leave in buckets of `dist.grad_bucket_blocks` consecutive blocks (one contiguous slice
            # of the stacked gradient): fewer, larger messages for the per-link-bound xGMI rings.
            deferred = None
            sent_from = self.n_layers  # blocks [sent_from, L) are on their way

            def dw_and_send(j):
                nonlocal sent_from
                be.fc_bwd_block_dw(buf, j)
                if sent_from - j >= dist.grad_bucket_blocks or j == 0:
                    dist.reduce_gradient_async(self.stacked_grad["w"][j:sent_from])
                    sent_from = j

            for i in range(self.n_layers - 1, -1, -1):
                be.fc_bwd_block_reduce(buf, i)
                h = dist.all_reduce_async(buf.bn_sums) if dist.sync_bn else None
                if deferred is not None:
                    dw_and_send(deferred)
                if h is not None:
                    h.wait()
                be.fc_bwd_block_dx(buf, i, gb if dist.sync_bn else buf.B)
                deferred = i
            if deferred is not None:
                dw_and_send(deferred)
            be.fc_bwd_end(buf)
            dist.reduce_gradient_async(self.flat_grads, exclude=self.stacked_grad["w"])
        for p, g in zip(self._param_list, self._grad_views):
            if p.grad is not g:
                p.grad = g

    def forward(self, x, depth=None, objective=None):
        """act_vec [depth+1, B, O].  ``objective`` (probability.HeadObjective, regression with one output) additionally
        runs the likelihood pass inside the head kernels and leaves its results in that object."""
        depth = self.n_layers if depth is None else depth
        if x.dim() == 2 and x.shape[0] == 0 and x.shape[1] == self.input_dim and not self.training:
            return x.new_zeros(depth + 1, 0, self.output_dim, dtype=torch.float32)  # (no rows: nothing to launch)
        if self.training and torch.is_grad_enabled():
            return _FcForward.apply(self, x, depth, objective, *self._param_list)
        return self._run_forward(x, depth, self.training, objective).act

    forward_get_acts = forward


class arq_uncert_fc_MLP(arq_uncert_fc_resnet):
    residual = False
    _block_cls = bern_MLPBlock_nores
