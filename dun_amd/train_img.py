"""Epoch loop of the image DUNs (reference: experiments/image/train_DUN.py:83-248), over loaders the caller supplies
(the corpus loaders -- torchvision downloads -- are out of scope).  Same bookkeeping as the
reference's train_loop: batch-size-weighted epoch means of (ELBO estimate, predictive NLL, error), q(d) warm-up
(``q_nograd_its`` epochs with q_logits frozen), MultiStepLR via ``net.update_lr()``, a dev pass per epoch, a checkpoint
per epoch plus ``model_best.pth.tar``, and the running statistics in ``stats_array.pkl``.
"""
import os
import pickle
import shutil
import time

import numpy as np
import torch

from dun_amd.probability import depth_categorical_VI
from dun_amd.stochastic_img_resnets import resnet18, resnet34, resnet50, resnet101
from dun_amd.training_wrappers import DUN_VI
from dun_amd.utils import cprint, mkdir

MODELS = {"resnet18": resnet18, "resnet34": resnet34, "resnet50": resnet50, "resnet101": resnet101}


def make_image_dun(dataset="CIFAR10", model="resnet50", start_depth=1, end_depth=13, lr=0.1, momentum=0.9,
                   weight_decay=1e-4, milestones=None, N_train=0, cuda=None):
    """The net train_DUN.py's main() builds (train_DUN.py:103-139): depth-uniform prior over end_depth - start_depth
    exits, ResNet DUN with zero-initialised residuals, DUN_VI wrapper with SGD and a MultiStepLR schedule."""
    if model not in MODELS:
        raise Exception("requested model not implemented")
    cuda = torch.cuda.is_available() if cuda is None else cuda
    initial_conv = "3x3" if dataset in ("Imagenet", "SmallImagenet") else "1x3"
    input_chanels = 1 if dataset in ("MNIST", "Fashion") else 3
    num_classes = 1000 if dataset in ("Imagenet", "SmallImagenet") else 100 if dataset == "CIFAR100" else 10
    n_layers = end_depth - start_depth
    prob_model = depth_categorical_VI([1 / n_layers] * n_layers, cuda=cuda)
    net_model = MODELS[model](arch_uncert=True, start_depth=start_depth, end_depth=end_depth, num_classes=num_classes,
                              zero_init_residual=True, initial_conv=initial_conv, concat_pool=False,
                              input_chanels=input_chanels, p_drop=0)
    return DUN_VI(net_model, prob_model, N_train, lr=lr, momentum=momentum, weight_decay=weight_decay, cuda=cuda,
                  schedule=milestones, regression=False, pred_sig=None)


def train_loop(net, train_loader, val_loader, Ntrain, epochs=90, resume="", savedir="./", save_all_epochs=False,
               q_nograd_its=0, best_err1=1.0):
    """Returns (marginal_loglike, err_train, train_loss, err_dev, dev_loss, best_err1), one entry per epoch."""
    mkdir(savedir)
    net.N_train = Ntrain
    start_epoch = 0
    marginal_loglike, train_loss, dev_loss = np.zeros(epochs), np.zeros(epochs), np.zeros(epochs)
    err_train, err_dev = np.zeros(epochs), np.zeros(epochs)
    stats_file = os.path.join(savedir, "stats_array.pkl")
    if resume:
        if os.path.isfile(resume):
            loaded = net.load(resume)
            start_epoch, best_err1 = loaded if isinstance(loaded, tuple) else (loaded, best_err1)
            print("=> loaded checkpoint '{}' (epoch {})".format(resume, start_epoch))
            progress = os.path.join(os.path.dirname(resume), "stats_array.pkl")
            if os.path.isfile(progress):
                with open(progress, "rb") as f:  # written by this loop below
                    loaded_stats = pickle.load(f)
                # (a run resumed with a longer schedule than the one that wrote the file keeps what it has)
                marginal_loglike, err_train, train_loss, err_dev, dev_loss = [
                    np.concatenate([a[:epochs], np.zeros(max(0, epochs - len(a)))]) for a in loaded_stats]
        else:
            print("=> no checkpoint found at '{}'".format(resume))
    if q_nograd_its > 0:
        net.prob_model.q_logits.requires_grad = False
    for epoch in range(start_epoch, epochs):
        if q_nograd_its > 0 and epoch == q_nograd_its:
            net.prob_model.q_logits.requires_grad = True
        tic = time.time()
        nb_samples = 0
        for x, y in train_loader:
            marg_loglike_estimate, minus_loglike, err = net.fit(x, y)
            marginal_loglike[epoch] += marg_loglike_estimate * x.shape[0]
            err_train[epoch] += err * x.shape[0]
            train_loss[epoch] += minus_loglike * x.shape[0]
            nb_samples += len(x)
        marginal_loglike[epoch] /= nb_samples
        train_loss[epoch] /= nb_samples
        err_train[epoch] /= nb_samples
        print("\n depth approx posterior", net.prob_model.current_posterior.data.cpu().numpy())
        print("it %d/%d, ELBO/evidence %.4f, pred minus loglike = %f, err = %f" %
              (epoch, epochs, marginal_loglike[epoch], train_loss[epoch], err_train[epoch]), end="")
        cprint("r", "   time: %f seconds\n" % (time.time() - tic))
        net.update_lr()
        tic = time.time()
        nb_samples = 0
        for x, y in val_loader:
            minus_loglike, err = net.eval(x, y)
            dev_loss[epoch] += minus_loglike * x.shape[0]
            err_dev[epoch] += err * x.shape[0]
            nb_samples += len(x)
        dev_loss[epoch] /= nb_samples
        err_dev[epoch] /= nb_samples
        cprint("g", "     pred minus loglike = %f, err = %f\n" % (dev_loss[epoch], err_dev[epoch]), end="")
        cprint("g", "    time: %f seconds\n" % (time.time() - tic))
        filename = "checkpoint.pth.tar"
        if save_all_epochs:
            filename = str(epoch) + "_" + filename
        net.save(os.path.join(savedir, filename), best_err1)
        if err_dev[epoch] < best_err1:
            best_err1 = err_dev[epoch]
            cprint("b", "best top1 dev err: %f" % err_dev[epoch])
            shutil.copyfile(os.path.join(savedir, filename), os.path.join(savedir, "model_best.pth.tar"))
        with open(stats_file, "wb") as f:
            pickle.dump([marginal_loglike, err_train, train_loss, err_dev, dev_loss], f)
    return marginal_loglike, err_train, train_loss, err_dev, dev_loss, best_err1
