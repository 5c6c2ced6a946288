"""Fused SGD(momentum, weight_decay): one kernel over each flat parameter buffer.

Numerics are torch.optim.SGD's (dampening 0, no nesterov) as the reference configures it at
src/DUN/training_wrappers.py:48-52 -- including weight decay on q_logits and log_std, which share the single
parameter group.  Parameters that live in a model's flat buffer (dun_amd.stochastic_fc_models) are stepped with
ONE segment for the whole buffer; every other parameter (q_logits, log_std, the conv trunk of image DUNs) is a
segment of its own, and up to 24 segments share a launch (dun_sgd_step_multi): three tensors = one launch for the
FC nets, a handful of launches instead of ~160 for a ResNet-50 DUN.  A parameter whose ``.grad`` is None is
skipped, like torch does (q warm-up).
"""
import torch

from . import kernels


def _dense_views(p, g, st):
    """(param, grad, momentum) as contiguous tensors over the SAME element order.  Contiguous parameters pass through;
    channels_last (NHWC) conv weights are stepped in their own memory order (an elementwise update does not care)."""
    if p.is_contiguous():
        if "momentum_buffer" not in st:
            st["momentum_buffer"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
        return p, (g if g.is_contiguous() else g.contiguous()), st["momentum_buffer"]
    if p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last):
        if "momentum_buffer" not in st or st["momentum_buffer"].stride() != p.stride():
            old = st.get("momentum_buffer")
            st["momentum_buffer"] = torch.zeros_like(p, memory_format=torch.channels_last)
            if old is not None:
                st["momentum_buffer"].copy_(old)
        if g.stride() != p.stride():
            g = g.contiguous(memory_format=torch.channels_last)
        return tuple(t.permute(0, 2, 3, 1) for t in (p, g, st["momentum_buffer"]))
    raise RuntimeError("FusedSGD: parameter with strides %s is neither contiguous nor channels_last" % (p.stride(),))


class FusedSGD(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-2, momentum=0.0, weight_decay=0.0):
        defaults = dict(lr=lr, momentum=momentum, weight_decay=weight_decay)
        super(FusedSGD, self).__init__(params, defaults)

    def zero_grad(self, set_to_none=True):
        # flat-buffer gradients are overwritten by every backward and must stay attached to their buffer
        for group in self.param_groups:
            for p in group["params"]:
                if getattr(p, "_dun_flat_owner", None) is None:
                    if set_to_none:
                        p.grad = None
                    elif p.grad is not None:
                        p.grad.zero_()

    @torch.no_grad()
    def step(self, closure=None, gate=None):
        """``gate``: a one-element tensor counting the batch's invalid labels (``backend.label_gate``); when it is
        positive nothing is updated -- parameters and momentum stay as they were -- like nn.CrossEntropyLoss raising
        before the update in the reference.  On the device the kernels read it themselves: no host synchronisation."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        be = kernels.backend()
        multi = getattr(be, "sgd_step_multi", None)  # several tensors per launch (the HIP backend has it)
        for group in self.param_groups:
            lr, mom, wd = group["lr"], group["momentum"], group["weight_decay"]
            owners, triples = [], []
            for p in group["params"]:
                owner = getattr(p, "_dun_flat_owner", None)
                if owner is not None:
                    if not any(o is owner for o in owners):
                        owners.append(owner)
                    continue
                if p.grad is None:
                    continue
                triples.append(_dense_views(p.data, p.grad, self.state[p]))
            for owner in owners:
                st = self.state[owner._param_list[0]]
                buf = st.get("flat_momentum_buffer")
                if buf is None or buf.shape != owner.flat_params.shape or buf.device != owner.flat_params.device:
                    buf = torch.zeros_like(owner.flat_params)
                    st["flat_momentum_buffer"] = buf
                triples.append((owner.flat_params, owner.flat_grads, buf))
            if multi is not None and (len(triples) > 1 or (gate is not None and triples)):
                multi(triples, lr, mom, wd) if gate is None else multi(triples, lr, mom, wd, gate=gate)
            elif gate is not None and float(gate) > 0:
                pass  # (host backends of the tests: the count is already at hand)
            else:
                for pv, gv, bv in triples:
                    be.sgd_step(pv, gv, bv, lr, mom, wd)
        return loss
