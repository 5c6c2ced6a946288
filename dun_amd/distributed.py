"""Batch-sharded data parallelism for the DUN path: one process per GPU, torch.distributed (RCCL over xGMI).

The reference has no distributed code; this is the sharding the north-star asks for.  Samples
are independent through Linear / ReLU / head / likelihood, so rank r just takes its rows.  What couples samples
is exchanged explicitly, and nothing else is:
  * BatchNorm batch statistics: fp64 [2,W] sums per block, forward and backward (``sync_bn=True`` reproduces the
    single-device numerics of the full batch; ``False`` gives per-replica statistics like DataParallel would);
    image DUNs: the same for every BatchNorm2d call of the convolutional trunk, fp64 [2,C] per call (sync_bn.py);
  * the per-depth likelihood sums: one fp64 [D] all-reduce per step (the VI backward does not wait for it);
  * parameter gradients: SUM all-reduce (the loss already carries 1/global_batch), issued per bucket of two blocks
    while the backward of the shallower blocks is still running, so the 2 x 16 MB messages overlap with the GEMMs.
Everything travels on ONE communicator (collectives of one communicator execute in issue order on one stream; several
communicators running concurrently are a known deadlock hazard when devices schedule their kernels in different
orders).  In the trunk backward the BatchNorm exchange of block i-1 is asynchronous and the weight-gradient GEMM of
block i runs underneath it (see arq_uncert_fc_resnet._run_backward); the forward exchanges are synchronous (nothing
worth hiding under them).  The gradient message that block i completes (two blocks, 33 MB) is issued right behind
that small backward exchange, so the NEXT small message -- which the compute stream will wait for -- is a dX GEMM,
a BatchNorm reduction and a dW GEMM away from it.
All ranks must hold the same number of rows (global batch = local batch * world size).
"""
import torch
import torch.distributed as dist


class Shard(object):
    def __init__(self, group=None, sync_bn=True, overlap=True, grad_bucket_blocks=2):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.sync_bn = sync_bn
        self.overlap = overlap
        self._pending = []
        self.grad_group = group
        self.grad_bucket_blocks = max(1, int(grad_bucket_blocks))  # blocks per weight-gradient message (W*W floats each)
        # forward BatchNorm exchange asynchronous too (only the 12 us per-depth head can run underneath it, less than
        # the two extra stream hand-offs cost on a one-rank communicator: off unless measured otherwise on 8 GPUs)
        self.async_forward_bn = False
        # rehearsal switch (scripts/shard_rehearsal.py): issue every collective even in a one-rank group, so the
        # stream hand-offs, async handles and message order of the N-GPU step run on a one-GPU box
        self.force_collectives = False

    def global_batch(self, local_batch):
        return local_batch * self.world

    def all_reduce(self, t):
        if self.world > 1 or self.force_collectives:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def all_reduce_async(self, t):
        """SUM all-reduce that the compute stream does not wait for yet: returns a handle (`.wait()` makes the current
        stream wait for the result) or None when there is nothing to exchange."""
        if self.world > 1 or self.force_collectives:
            return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        return None

    def reduce_gradient_async(self, t, exclude=None):
        """SUM all-reduce of a gradient slice; `exclude` is a sub-range of `t`'s storage that was already sent."""
        if self.world == 1 and not self.force_collectives:
            return
        pieces = [t]
        if exclude is not None:
            flat = t.reshape(-1)
            lo = exclude.storage_offset() - t.storage_offset()
            hi = lo + exclude.numel()
            pieces = [p for p in (flat[:lo], flat[hi:]) if p.numel()]
        for p in pieces:
            if self.overlap:
                self._pending.append(dist.all_reduce(p, op=dist.ReduceOp.SUM, group=self.grad_group, async_op=True))
            else:
                dist.all_reduce(p, op=dist.ReduceOp.SUM, group=self.grad_group)

    def wait_gradients(self):
        for h in self._pending:
            h.wait()
        self._pending = []


def shard(net, group=None, sync_bn=True, overlap=True, grad_bucket_blocks=2):
    """Make a DUN / DUN_VI wrapper train on this rank's shard of every global batch."""
    spec = Shard(group, sync_bn=sync_bn, overlap=overlap, grad_bucket_blocks=grad_bucket_blocks)
    # replicas must start identical: rank 0's parameters, BatchNorm buffers, q_logits and log_std win
    with torch.no_grad():
        tensors = [p.data for p in net.model.parameters() if getattr(p, "_dun_flat_owner", None) is None]
        tensors += [b for b in net.model.buffers() if getattr(net.model, "flat_params", None) is None]
        if getattr(net.model, "flat_params", None) is not None:
            tensors += [net.model.flat_params, net.model.running_mean, net.model.running_var,
                        net.model.num_batches_tracked]
        tensors += [p.data for p in net.prob_model.parameters()]
        tensors += [p.data for p in net.f_neg_loglike.parameters()]
        if spec.world > 1:
            for t in tensors:
                dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    # image DUNs: every BatchNorm2d of the conv trunk normalises with the statistics of the global batch (sync_bn.py)
    spec.sync_bn2d_modules = 0
    if spec.sync_bn:
        from .sync_bn import convert
        spec.sync_bn2d_modules = convert(net.model, spec)
    net.model._dist = spec
    net.prob_model._dist = spec
    net._dist = spec
    return spec
