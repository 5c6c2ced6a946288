"""Epoch loop for FC DUNs with the reference's signature and return tuple (src/DUN/train_fc.py:12-148).

Per epoch: ``net.fit`` over the train loader with batch-size-weighted accumulation of (objective, NLL, err),
optional exact-posterior / exact-ELBO tracking, a dev pass every ``nb_its_dev`` epochs, `theta_last.dat` /
`theta_best.dat` checkpoints, early stopping on the running objective, and the q warm-up used by the scripts
(``q_nograd_its`` epochs with ``q_logits`` frozen).
"""
import os
import tempfile
import time

import numpy as np
import torch

from .utils import cprint, mkdir


def _weighted_pass(loader, step, flat_ims, n_out):
    tot = np.zeros(n_out)
    n = 0
    for x, y in loader:
        if flat_ims:
            x = x.view(x.shape[0], -1)
        out = step(x, y)
        tot += np.asarray(out, dtype=np.float64) * x.shape[0]
        n += len(x)
    return tot / max(n, 1)


def train_fc_DUN(net, name, save_dir, batch_size, nb_epochs, train_loader, val_loader,
                 cuda, seed, flat_ims=False, nb_its_dev=1, early_stop=None,
                 track_posterior=False, track_exact_ELBO=False, tags=None,
                 load_path=None, save_freq=None, q_nograd_its=0):
    basedir = os.path.join(save_dir, name, next(tempfile._get_candidate_names()))
    media_dir = basedir + '/media/'
    models_dir = basedir + '/models/'
    mkdir(models_dir)
    mkdir(media_dir)

    if seed is not None:
        torch.manual_seed(seed)
        if cuda:
            torch.cuda.manual_seed(seed)

    marginal_loglike_estimate = np.zeros(nb_epochs)
    train_mean_predictive_loglike = np.zeros(nb_epochs)
    dev_mean_predictive_loglike = np.zeros(nb_epochs)
    err_train = np.zeros(nb_epochs)
    err_dev = np.zeros(nb_epochs)
    true_d_posterior, approx_d_posterior, true_likelihood, exact_ELBO = [], [], [], []

    best_epoch = 0
    best_marginal_loglike = -np.inf
    if q_nograd_its > 0:
        net.prob_model.q_logits.requires_grad = False

    tic0 = time.time()
    i = -1
    for i in range(nb_epochs):
        if q_nograd_its > 0 and i == q_nograd_its:
            net.prob_model.q_logits.requires_grad = True

        net.set_mode_train(True)
        tic = time.time()
        marginal_loglike_estimate[i], train_mean_predictive_loglike[i], err_train[i] = \
            _weighted_pass(train_loader, net.fit, flat_ims, 3)
        toc = time.time()

        print('\n depth approx posterior', net.prob_model.current_posterior.data.cpu().numpy())
        print("it %d/%d, ELBO/evidence %.4f, pred minus loglike = %f, err = %f" %
              (i, nb_epochs, marginal_loglike_estimate[i], train_mean_predictive_loglike[i], err_train[i]), end="")
        cprint('r', '   time: %f seconds\n' % (toc - tic))
        net.update_lr()

        if track_posterior:
            approx_d_posterior.append(net.prob_model.current_posterior.data.cpu().numpy())
            exact_posterior, log_marginal_over_depth = net.get_exact_d_posterior(train_loader, train_bn=True,
                                                                                 logposterior=False)
            true_d_posterior.append(exact_posterior.data.cpu().numpy())
            true_likelihood.append(log_marginal_over_depth)
        if track_exact_ELBO:
            exact_ELBO.append(net.get_exact_ELBO(train_loader, train_bn=True))

        if i % nb_its_dev == 0:
            tic = time.time()
            dev_mean_predictive_loglike[i], err_dev[i] = _weighted_pass(val_loader, net.eval, flat_ims, 2)
            toc = time.time()
            cprint('g', '     pred minus loglike = %f, err = %f\n' % (dev_mean_predictive_loglike[i], err_dev[i]), end="")
            cprint('g', '    time: %f seconds\n' % (toc - tic))

        if save_freq is not None and i % save_freq == 0:
            net.save(models_dir + '/theta_last.dat')

        if marginal_loglike_estimate[i] > best_marginal_loglike:
            best_marginal_loglike = marginal_loglike_estimate[i]
            best_epoch = i
            cprint('b', 'best marginal loglike: %f' % best_marginal_loglike)
            if i % 2 == 0:
                net.save(models_dir + '/theta_best.dat')

        if early_stop is not None and (i - best_epoch) > early_stop:
            cprint('r', '   stopped early!\n')
            break

    runtime_per_it = (time.time() - tic0) / float(i + 1)
    cprint('r', '   average time: %f seconds\n' % runtime_per_it)

    if track_posterior:
        approx_d_posterior = np.stack(approx_d_posterior, axis=0)
        true_d_posterior = np.stack(true_d_posterior, axis=0)
        true_likelihood = np.stack(true_likelihood, axis=0)
    if track_exact_ELBO:
        exact_ELBO = np.stack(exact_ELBO, axis=0)

    return marginal_loglike_estimate, train_mean_predictive_loglike, dev_mean_predictive_loglike, err_train, err_dev, \
        approx_d_posterior, true_d_posterior, true_likelihood, exact_ELBO, basedir
