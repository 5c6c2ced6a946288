"""Synchronised BatchNorm2d for the image DUN's convolutional trunk under ``shard(sync_bn=True)``.

The reference trains image DUNs on one device, where every BatchNorm2d call -- trunk blocks, the 1x1 adapt stacks and
the tail that each exit runs again (src/DUN/stochastic_img_resnets.py:280-286, 304-315) -- normalises with the
statistics of the WHOLE batch.  With the batch sharded over ranks that needs one exchange per call, forward and
backward; torch's ``nn.SyncBatchNorm`` only runs on GPUs with NCCL/RCCL, so this is a small autograd Function that works
with any backend (RCCL on the GPUs, gloo in the CPU tests) and rides the single communicator of ``Shard``:

  forward   every rank computes its local (mean_r, biased var_r) with torch's var_mean, then ONE fp64 [2,C] SUM
            all-reduce of (n_r mean_r, n_r (var_r + mean_r^2)) gives the global mean and the global E[x^2] -- the
            parallel-variance combination; the squares are taken of MEANS in fp64, so the subtraction loses nothing;
  backward  ONE fp64 [2,C] SUM all-reduce of (sum dy, sum dy xhat); dx uses the global sums, dweight / dbias are the
            LOCAL sums: parameter gradients are summed over ranks by the wrapper's gradient all-reduce like all others.

All ranks hold the same number of rows (Shard's contract), so the global count is local count x world size.
Running statistics are updated from the global batch (unbiased variance), as a single device would.
"""
import torch
import torch.nn as nn


class _SyncBN2dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, eps, momentum, spec):
        C = x.shape[1]
        dims = (0, 2, 3)
        n_local = x.numel() // C
        n = spec.global_batch(n_local)
        var_l, mean_l = torch.var_mean(x.detach(), dim=dims, unbiased=False)
        m64 = mean_l.double()
        msg = torch.stack([m64 * n_local, (var_l.double() + m64 * m64) * n_local])
        spec.all_reduce(msg)
        mean64 = msg[0] / n
        var64 = (msg[1] / n - mean64 * mean64).clamp_(min=0.0)
        mean = mean64.to(x.dtype)
        rstd = torch.rsqrt(var64 + eps).to(x.dtype)
        if running_mean is not None:
            with torch.no_grad():
                unbiased = var64 * (n / max(n - 1.0, 1.0))
                running_mean.mul_(1.0 - momentum).add_(mean64.to(running_mean.dtype), alpha=momentum)
                running_var.mul_(1.0 - momentum).add_(unbiased.to(running_var.dtype), alpha=momentum)
        xhat = (x.detach() - mean.view(1, C, 1, 1)) * rstd.view(1, C, 1, 1)
        y = xhat * weight.detach().view(1, C, 1, 1) + bias.detach().view(1, C, 1, 1)
        ctx.save_for_backward(xhat, weight, rstd)
        ctx.spec, ctx.n = spec, n
        return y

    @staticmethod
    def backward(ctx, dy):
        xhat, weight, rstd = ctx.saved_tensors
        C = xhat.shape[1]
        dims = (0, 2, 3)
        sums = torch.stack([dy.sum(dim=dims, dtype=torch.float64), (dy * xhat).sum(dim=dims, dtype=torch.float64)])
        dbias = sums[0].to(dy.dtype)
        dweight = sums[1].to(dy.dtype)     # local sums: the wrapper's gradient all-reduce adds the ranks up
        ctx.spec.all_reduce(sums)
        m = (sums / ctx.n).to(dy.dtype)
        dx = (dy - m[0].view(1, C, 1, 1) - xhat * m[1].view(1, C, 1, 1)) * (weight.detach() * rstd).view(1, C, 1, 1)
        return dx, dweight, dbias, None, None, None, None, None


class SyncBatchNorm2d(nn.BatchNorm2d):
    """nn.BatchNorm2d whose train-mode statistics are those of the global batch.  Same parameters, buffers and
    state_dict keys: ``convert`` re-classes existing modules in place, nothing is copied."""

    _dun_dist = None

    def __getstate__(self):
        # a pickled model (BaseNet.save) does not carry its communicator: loaded again it is a plain BatchNorm2d until
        # shard() binds it to the process group of the job that loaded it
        st = self.__dict__.copy()
        st["_dun_dist"] = None
        return st

    def forward(self, x):
        spec = self._dun_dist
        if not self.training or spec is None or (spec.world == 1 and not spec.force_collectives) or not spec.sync_bn:
            return super(SyncBatchNorm2d, self).forward(x)
        if not self.track_running_stats or not self.affine or self.momentum is None:
            raise NotImplementedError("SyncBatchNorm2d: affine, tracked running statistics and a fixed momentum only "
                                      "(what the DUN ResNets construct)")
        self._check_input_dim(x)
        self.num_batches_tracked.add_(1)
        return _SyncBN2dFn.apply(x, self.weight, self.bias, self.running_mean, self.running_var, self.eps,
                                 self.momentum, spec)


def convert(model, spec):
    """Every nn.BatchNorm2d of `model` becomes a SyncBatchNorm2d bound to `spec`; returns how many."""
    k = 0
    for m in model.modules():
        if type(m) is nn.BatchNorm2d or type(m) is SyncBatchNorm2d:
            m.__class__ = SyncBatchNorm2d
            m._dun_dist = spec
            k += 1
    return k
