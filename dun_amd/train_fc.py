"""Epoch loop for FC DUNs: same call signature, side effects and return tuple as the reference's ``train_fc_DUN``
(src/DUN/train_fc.py:12-148), organised around a small per-run record instead of a dozen parallel arrays.

What an epoch does, in order: q warm-up switch (``q_nograd_its`` epochs with ``q_logits`` frozen) -> one pass of
``net.fit`` over the train loader (batch-size-weighted means of objective / NLL / error) -> ``net.update_lr()`` ->
optional exact depth posterior and exact ELBO over the whole train set -> a dev pass every ``nb_its_dev`` epochs ->
``theta_last.dat`` every ``save_freq`` epochs, ``theta_best.dat`` when the running objective improves on an even epoch
-> early stop when it has not improved for ``early_stop`` epochs.
"""
import os
import tempfile
import time

import numpy as np
import torch

from ._epochs import QWarmup, epoch_means
from .utils import cprint, mkdir


class _RunLog(object):
    """Per-epoch curves of one run; `as_reference_tuple` lays them out the way the reference returns them."""

    def __init__(self, epochs):
        self.objective = np.zeros(epochs)
        self.train_nll = np.zeros(epochs)
        self.dev_nll = np.zeros(epochs)
        self.train_err = np.zeros(epochs)
        self.dev_err = np.zeros(epochs)
        self.q, self.posterior, self.log_marginal, self.exact_elbo = [], [], [], []

    @staticmethod
    def _stack(rows):
        return np.stack(rows, axis=0) if rows else rows

    def as_reference_tuple(self, track_posterior, track_exact_elbo, basedir):
        q, post, marg = ((self._stack(self.q), self._stack(self.posterior), self._stack(self.log_marginal))
                         if track_posterior else (self.q, self.posterior, self.log_marginal))
        exact = self._stack(self.exact_elbo) if track_exact_elbo else self.exact_elbo
        return (self.objective, self.train_nll, self.dev_nll, self.train_err, self.dev_err, q, post, marg, exact,
                basedir)


def train_fc_DUN(net, name, save_dir, batch_size, nb_epochs, train_loader, val_loader,
                 cuda, seed, flat_ims=False, nb_its_dev=1, early_stop=None,
                 track_posterior=False, track_exact_ELBO=False, tags=None,
                 load_path=None, save_freq=None, q_nograd_its=0):
    basedir = os.path.join(save_dir, name, next(tempfile._get_candidate_names()))
    models_dir = os.path.join(basedir, "models")
    for d in (models_dir, os.path.join(basedir, "media")):
        mkdir(d + "/")
    if seed is not None:
        torch.manual_seed(seed)
        if cuda:
            torch.cuda.manual_seed(seed)

    log = _RunLog(nb_epochs)
    warmup = QWarmup(net.prob_model, q_nograd_its)
    best_value, best_epoch, epochs_run = -np.inf, 0, 0
    started = time.time()

    for ep in range(nb_epochs):
        epochs_run = ep + 1
        warmup.at_epoch(ep)

        net.set_mode_train(True)
        t0 = time.time()
        log.objective[ep], log.train_nll[ep], log.train_err[ep] = epoch_means(train_loader, net.fit, 3, flat_ims)
        q_now = net.prob_model.current_posterior.data.cpu().numpy()
        print("\n depth approx posterior", q_now)
        print("it %d/%d, ELBO/evidence %.4f, pred minus loglike = %f, err = %f"
              % (ep, nb_epochs, log.objective[ep], log.train_nll[ep], log.train_err[ep]), end="")
        cprint("r", "   time: %f seconds\n" % (time.time() - t0))
        net.update_lr()

        if track_posterior:
            posterior, log_marginal = net.get_exact_d_posterior(train_loader, train_bn=True, logposterior=False)
            log.q.append(q_now)
            log.posterior.append(posterior.data.cpu().numpy())
            log.log_marginal.append(log_marginal)
        if track_exact_ELBO:
            log.exact_elbo.append(net.get_exact_ELBO(train_loader, train_bn=True))

        if ep % nb_its_dev == 0:
            t0 = time.time()
            log.dev_nll[ep], log.dev_err[ep] = epoch_means(val_loader, net.eval, 2, flat_ims)
            cprint("g", "     pred minus loglike = %f, err = %f\n" % (log.dev_nll[ep], log.dev_err[ep]), end="")
            cprint("g", "    time: %f seconds\n" % (time.time() - t0))

        if save_freq is not None and ep % save_freq == 0:
            net.save(os.path.join(models_dir, "theta_last.dat"))
        if log.objective[ep] > best_value:
            best_value, best_epoch = log.objective[ep], ep
            cprint("b", "best marginal loglike: %f" % best_value)
            if ep % 2 == 0:
                net.save(os.path.join(models_dir, "theta_best.dat"))
        if early_stop is not None and ep - best_epoch > early_stop:
            cprint("r", "   stopped early!\n")
            break

    cprint("r", "   average time: %f seconds\n" % ((time.time() - started) / float(max(epochs_run, 1))))
    return log.as_reference_tuple(track_posterior, track_exact_ELBO, basedir)
