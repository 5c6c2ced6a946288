"""Host glue shared by the wrappers; mirrors the names the reference's callers use from src/utils.py."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F
from torch.optim.lr_scheduler import MultiStepLR


def mkdir(paths):
    if not isinstance(paths, (list, tuple)):
        paths = [paths]
    for path in paths:
        os.makedirs(path, mode=0o777, exist_ok=True)


_COLORS = dict(a="30", r="31", g="32", y="33", b="34", p="35", c="36", w="37")


def cprint(color, text, **kwargs):
    """ANSI-coloured print, '*x' for bold (same call convention as src/utils.py:24-41)."""
    bold = color.startswith("*")
    code = _COLORS[color[1:] if bold else color]
    print("\x1b[%s%sm%s\x1b[0m" % ("1;" if bold else "", code, text), **kwargs)
    sys.stdout.flush()


def to_variable(var=(), cuda=True, volatile=False):
    """numpy -> float32 tensor, and onto the GPU when cuda (src/utils.py:49-59). `volatile` is accepted and ignored."""
    out = []
    for v in var:
        if isinstance(v, np.ndarray):
            v = torch.from_numpy(v).type(torch.FloatTensor)
        if cuda and not v.is_cuda:
            v = v.cuda(non_blocking=True)
        out.append(v)
    return out


def rms(x, y):
    return F.mse_loss(x, y, reduction="mean").sqrt()


class BaseNet(object):
    """Wrapper base (src/utils.py:102-182): parameter count, mode switch, LR schedule, whole-object save/load."""

    def __init__(self):
        cprint("c", "\nNet:")
        self.scheduler = None

    def get_nb_parameters(self):
        return int(sum(p.numel() for p in self.model.parameters()))

    def set_mode_train(self, train=True):
        if train:
            self.model.train()
        else:
            self.model.eval()

    def make_scheduler(self, gamma=0.1, milestones=None):
        self.scheduler = MultiStepLR(self.optimizer, milestones=milestones, gamma=gamma)

    def update_lr(self):
        self.epoch += 1
        if self.scheduler is not None:
            self.scheduler.step()

    def __getstate__(self):
        st = self.__dict__.copy()
        for k in ("_graph_state", "_graph_states", "_metrics_out", "_metrics_signalled"):  # captured hipGraphs do not pickle
            st.pop(k, None)
        st["_dist"] = None
        return st

    def save(self, filename, best_err=None):
        cprint("c", "Writting %s\n" % filename)
        state = {"epoch": self.epoch, "lr": self.lr, "model": self.model, "prob_model": self.prob_model,
                 "f_neg_loglike": self.f_neg_loglike, "optimizer": self.optimizer, "scheduler": self.scheduler}
        if best_err is not None:
            state["best_err"] = best_err
        torch.save(state, filename)

    def load(self, filename, parallel=False, to_cpu=False):
        kw = dict(weights_only=False)  # files written by save() above (whole objects), never reference files
        if to_cpu:
            kw["map_location"] = lambda storage, loc: storage
        state = torch.load(filename, **kw)
        self.epoch = state["epoch"]
        self.lr = state["lr"]
        self.model = state["model"]
        if "prob_model" in state:
            self.prob_model = state["prob_model"]
        self.f_neg_loglike = state["f_neg_loglike"]
        self.optimizer = state["optimizer"]
        self.scheduler = state["scheduler"]
        self.__dict__.pop("_graph_state", None)
        self.__dict__.pop("_graph_states", None)
        print("  restoring epoch: %d, lr: %f" % (self.epoch, self.lr))
        if "best_err" in state:
            return self.epoch, state["best_err"]
        return self.epoch

    def save_state(self, filename, best_err=None):
        """Tensors-only checkpoint (dun_amd.checkpoint): state dicts instead of pickled objects."""
        from .checkpoint import save_state
        cprint("c", "Writting %s\n" % filename)
        save_state(self, filename, best_err)

    def load_state(self, filename, to_cpu=False):
        """Restore from save_state's file, or from a reference checkpoint converted by
        scripts/export_reference_checkpoint.py.  Executes nothing from the file."""
        from .checkpoint import load_state
        return load_state(self, filename, to_cpu=to_cpu)
