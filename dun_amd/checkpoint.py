"""Tensors-only checkpoints.

The reference's ``BaseNet.save`` (src/utils.py:124-154) pickles whole objects -- model, prob_model, likelihood, optimizer,
scheduler -- so its files are bound to the class paths of the code that wrote them and cannot be read without executing
them.  ``BaseNet.save / load`` of this package keep that behaviour for files this package wrote.  This module adds the
portable form: one dict of plain tensors, numbers, strings and lists (readable with ``torch.load(weights_only=True)``,
which executes nothing from the file), laid out as the STATE DICTS of the same five objects:

    {"format": "dun.state/1", "epoch", "lr", ["best_err"],
     "model":         model.state_dict()                       (the reference's keys: input_layer.weight, ...)
     "prob_model":    {"q_logits"?, "prior_probs", "current_posterior"?}
     "f_neg_loglike": {"log_std"?}
     "optimizer":     torch.optim.SGD.state_dict() layout: {"state": {i: {"momentum_buffer"}}, "param_groups": [...]}
                      over the reference's parameter order (model, prob_model, log_std: training_wrappers.py:48-52)
     "scheduler":     None or MultiStepLR's state with the milestone Counter flattened to a sorted list}

``net_state`` reads any wrapper with the reference's attribute names -- this package's DUN / DUN_VI or the reference's own
objects -- so a reference checkpoint is converted by loading it ONCE in the environment that wrote it
(scripts/export_reference_checkpoint.py) and read here ever after.  ``load_state`` puts such a file into a constructed
wrapper of this package, momentum included (per-parameter buffers are scattered into the flat momentum buffer the fused
SGD kernel works on).
"""
from collections import Counter, OrderedDict

import torch

FORMAT = "dun.state/1"
_PRIMITIVE = (bool, int, float, str, type(None))


def _plain(v):
    """Value restricted to what a weights-only load accepts (tensors, primitives, lists / dicts of those) or None."""
    if isinstance(v, torch.Tensor):
        return v.detach().clone().contiguous().cpu()
    if isinstance(v, _PRIMITIVE):
        return v
    if isinstance(v, (list, tuple)):
        out = [_plain(e) for e in v]
        return None if any(o is None and e is not None for o, e in zip(out, v)) else out
    return None


def _tensor_dict(sd):
    return OrderedDict((k, _plain(v)) for k, v in sd.items())


def _momentum_of(opt, p):
    """Momentum of one parameter as a dense tensor of the parameter's shape (None before the first step)."""
    owner = getattr(p, "_dun_flat_owner", None)
    if owner is None:
        buf = opt.state.get(p, {}).get("momentum_buffer")
        if buf is None:
            return None
        # channels_last conv weights keep their momentum in the same memory order: as a logical tensor it is p-shaped
        return buf.detach().clone().contiguous()
    flat = opt.state.get(owner._param_list[0], {}).get("flat_momentum_buffer")
    if flat is None:
        return None
    off = p.data.storage_offset() - owner.flat_params.storage_offset()
    return flat[off:off + p.numel()].view(p.shape).detach().clone()


def _optimizer_state(opt):
    state, groups, idx = {}, [], 0
    for g in opt.param_groups:
        ids = []
        for p in g["params"]:
            m = _momentum_of(opt, p)
            if m is not None:
                state[idx] = {"momentum_buffer": m.cpu()}
            ids.append(idx)
            idx += 1
        plain = {k: _plain(v) for k, v in g.items() if k != "params"}
        plain = {k: v for k, v in plain.items() if v is not None or g[k] is None}
        plain.setdefault("dampening", 0)
        plain.setdefault("nesterov", False)
        plain["params"] = ids
        groups.append(plain)
    return {"state": state, "param_groups": groups}


def _scheduler_state(sch):
    if sch is None:
        return None
    out = {}
    for k, v in sch.state_dict().items():
        if k == "milestones":
            out[k] = sorted(v.elements()) if isinstance(v, Counter) else sorted(v)
        else:
            v = _plain(v)
            if v is not None:
                out[k] = v
    return out


def net_state(net, best_err=None):
    """The wrapper's training state as a tensors-only dict (see the module docstring)."""
    pm = net.prob_model
    pm_state = {"prior_probs": _plain(torch.as_tensor(pm.prior_probs))}
    if getattr(pm, "q_logits", None) is not None:
        pm_state["q_logits"] = _plain(pm.q_logits)
    if getattr(pm, "current_posterior", None) is not None:
        pm_state["current_posterior"] = _plain(torch.as_tensor(pm.current_posterior))
    f = net.f_neg_loglike
    out = {"format": FORMAT, "epoch": int(net.epoch), "lr": float(net.lr),
           "model": _tensor_dict(net.model.state_dict()), "prob_model": pm_state,
           "f_neg_loglike": _tensor_dict(f.state_dict()) if isinstance(f, torch.nn.Module) else {},
           "optimizer": _optimizer_state(net.optimizer), "scheduler": _scheduler_state(getattr(net, "scheduler", None))}
    if best_err is not None:
        out["best_err"] = float(best_err)
    return out


def save_state(net, filename, best_err=None):
    torch.save(net_state(net, best_err), filename)


def _load_momentum(opt, saved):
    params = [p for g in opt.param_groups for p in g["params"]]
    n_saved = sum(len(g["params"]) for g in saved["param_groups"])
    if n_saved != len(params):
        raise ValueError("checkpoint optimizer covers %d parameters, this wrapper's has %d" % (n_saved, len(params)))
    for g, sg in zip(opt.param_groups, saved["param_groups"]):
        for k in ("lr", "momentum", "weight_decay", "initial_lr"):
            if k in sg:
                g[k] = sg[k]
        if sg.get("dampening", 0) != 0 or sg.get("nesterov", False):
            raise ValueError("checkpoint optimizer uses dampening / nesterov, which the fused SGD step does not implement")
    for idx, p in enumerate(params):
        st = saved["state"].get(idx, saved["state"].get(str(idx)))
        m = None if st is None else st.get("momentum_buffer")
        owner = getattr(p, "_dun_flat_owner", None)
        if owner is not None:
            ost = opt.state[owner._param_list[0]]
            flat = ost.get("flat_momentum_buffer")
            if flat is None or flat.shape != owner.flat_params.shape or flat.device != owner.flat_params.device:
                flat = ost["flat_momentum_buffer"] = torch.zeros_like(owner.flat_params)
            off = p.data.storage_offset() - owner.flat_params.storage_offset()
            dst = flat[off:off + p.numel()].view(p.shape)
            if m is None:
                dst.zero_()
            else:
                if tuple(m.shape) != tuple(p.shape):
                    raise ValueError("momentum of parameter %d has shape %s, expected %s" % (idx, tuple(m.shape), tuple(p.shape)))
                dst.copy_(m)
        elif m is None:
            opt.state[p].pop("momentum_buffer", None)
        else:
            if tuple(m.shape) != tuple(p.shape):
                raise ValueError("momentum of parameter %d has shape %s, expected %s" % (idx, tuple(m.shape), tuple(p.shape)))
            opt.state[p]["momentum_buffer"] = m.to(device=p.device, dtype=p.dtype).contiguous(
                memory_format=torch.channels_last if p.dim() == 4 and not p.is_contiguous() else torch.contiguous_format)


def _validate_shapes(net, st):
    """Every shape the loader is about to copy, checked before the first byte is overwritten: a checkpoint of the right
    parameter COUNT but another width / depth must leave the wrapper as it was."""
    here = net.model.state_dict()
    saved = st["model"]
    missing = [k for k in here if k not in saved]
    extra = [k for k in saved if k not in here]
    if missing or extra:
        raise ValueError("checkpoint model state does not match this model: missing %s, unexpected %s" % (missing[:3], extra[:3]))
    for k, v in here.items():
        if tuple(saved[k].shape) != tuple(v.shape):
            raise ValueError("checkpoint tensor %s has shape %s, this model's has %s" % (k, tuple(saved[k].shape), tuple(v.shape)))
    pm = net.prob_model
    if "q_logits" in st["prob_model"] and tuple(st["prob_model"]["q_logits"].shape) != tuple(pm.q_logits.shape):
        raise ValueError("checkpoint q_logits has shape %s, this prob_model's has %s"
                         % (tuple(st["prob_model"]["q_logits"].shape), tuple(pm.q_logits.shape)))
    if isinstance(net.f_neg_loglike, torch.nn.Module) and st["f_neg_loglike"]:
        for k, v in net.f_neg_loglike.state_dict().items():
            if k in st["f_neg_loglike"] and tuple(st["f_neg_loglike"][k].shape) != tuple(v.shape):
                raise ValueError("checkpoint likelihood tensor %s has shape %s, expected %s"
                                 % (k, tuple(st["f_neg_loglike"][k].shape), tuple(v.shape)))
    params = [p for g in net.optimizer.param_groups for p in g["params"]]
    for sg in st["optimizer"]["param_groups"]:
        if sg.get("dampening", 0) != 0 or sg.get("nesterov", False):
            raise ValueError("checkpoint optimizer uses dampening / nesterov, which the fused SGD step does not implement")
    for idx, p in enumerate(params):
        e = st["optimizer"]["state"].get(idx, st["optimizer"]["state"].get(str(idx)))
        m = None if e is None else e.get("momentum_buffer")
        if m is not None and tuple(m.shape) != tuple(p.shape):
            raise ValueError("momentum of parameter %d has shape %s, expected %s" % (idx, tuple(m.shape), tuple(p.shape)))


def load_state(net, filename_or_state, to_cpu=False):
    """Restore a constructed wrapper from ``save_state``'s file (or the dict itself).  -> epoch, or (epoch, best_err)
    like the reference's ``load`` (src/utils.py:158-182).  Nothing in the file is executed."""
    st = filename_or_state
    if not isinstance(st, dict):
        st = torch.load(st, weights_only=True, map_location="cpu" if to_cpu else None)
    if st.get("format") != FORMAT:
        raise ValueError("not a %s checkpoint (format = %r)" % (FORMAT, st.get("format")))
    pm = net.prob_model
    # refuse before anything is overwritten
    if "q_logits" in st["prob_model"] and getattr(pm, "q_logits", None) is None:
        raise ValueError("checkpoint holds q_logits but this wrapper's prob_model is the exact-posterior one")
    n_saved = sum(len(g["params"]) for g in st["optimizer"]["param_groups"])
    n_here = sum(len(g["params"]) for g in net.optimizer.param_groups)
    if n_saved != n_here:
        raise ValueError("checkpoint optimizer covers %d parameters, this wrapper's has %d" % (n_saved, n_here))
    _validate_shapes(net, st)
    net.model.load_state_dict(st["model"])
    with torch.no_grad():
        if "q_logits" in st["prob_model"]:
            pm.q_logits.data.copy_(st["prob_model"]["q_logits"])
        if st["prob_model"].get("current_posterior") is not None:
            ref = pm.q_logits if getattr(pm, "q_logits", None) is not None else torch.as_tensor(pm.prior_probs)
            pm.current_posterior = st["prob_model"]["current_posterior"].to(ref.device)
        if isinstance(net.f_neg_loglike, torch.nn.Module) and st["f_neg_loglike"]:
            net.f_neg_loglike.load_state_dict(st["f_neg_loglike"])
    _load_momentum(net.optimizer, st["optimizer"])
    sch = getattr(net, "scheduler", None)
    if st["scheduler"] is not None:
        if sch is None:
            net.make_scheduler(gamma=st["scheduler"].get("gamma", 0.1), milestones=list(st["scheduler"]["milestones"]))
            sch = net.scheduler
        sd = dict(st["scheduler"])
        sd["milestones"] = Counter(sd["milestones"])
        sch.load_state_dict(sd)
    net.epoch, net.lr = st["epoch"], st["lr"]
    net.__dict__.pop("_graph_states", None)  # captured steps were captured on the old buffers
    net.__dict__.pop("_graph_state", None)
    print("  restoring epoch: %d, lr: %f" % (net.epoch, net.lr))
    return (net.epoch, st["best_err"]) if "best_err" in st else net.epoch
