#!/usr/bin/env python
"""Config-1 plumbing (BASELINE.json configs[0]): DUN on the toy 'wiggle' set, the flow of the reference's
experiments/toy/train_toy.py:129-162 without the plotting.   python examples/train_toy.py --gpu 0 --n_epochs 200"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dun_amd.datasets import Datafeed, load_agw_1d, load_matern_1d, load_my_1d, load_wiggle  # noqa: E402
from dun_amd.probability import depth_categorical_VI  # noqa: E402
from dun_amd.stochastic_fc_models import arq_uncert_fc_MLP, arq_uncert_fc_resnet  # noqa: E402
from dun_amd.train_fc import train_fc_DUN  # noqa: E402
from dun_amd.training_wrappers import DUN_VI  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--n_epochs', type=int, default=4000)
ap.add_argument('--gpu', type=int, default=0)
ap.add_argument('--N_layers', type=int, default=10)
ap.add_argument('--width', type=int, default=100)
ap.add_argument('--lr', type=float, default=1e-3)
ap.add_argument('--wd', type=float, default=1e-4)
ap.add_argument('--network', default='ResNet', choices=['ResNet', 'MLP'])
ap.add_argument('--savedir', default='./saves/')
ap.add_argument('--dataset', default='wiggle', choices=['wiggle', 'agw_1d', 'matern_1d', 'my_1d'],
                help="the reference's --dataset values that need no download (train_toy.py:69-83)")
ap.add_argument('--datadir', default='./data/', help="a DUN checkout's experiments/data (the .npy toy sets)")
args = ap.parse_args()

torch.cuda.set_device(args.gpu)
if args.dataset == 'wiggle':
    X, Y = load_wiggle()
elif args.dataset == 'agw_1d':
    X, Y = load_agw_1d(args.datadir)
elif args.dataset == 'matern_1d':
    X, Y = load_matern_1d(args.datadir)
    X, Y = X.astype('float32'), Y.astype('float32')
else:
    X, Y = load_my_1d(args.datadir)[:2]
ds = Datafeed(torch.Tensor(X), torch.Tensor(Y))
train = torch.utils.data.DataLoader(ds, batch_size=2048, shuffle=True)
val = torch.utils.data.DataLoader(ds, batch_size=2048, shuffle=False)
ctor = arq_uncert_fc_resnet if args.network == 'ResNet' else arq_uncert_fc_MLP
model = ctor(X.shape[1], Y.shape[1], args.width, args.N_layers, w_prior=None, BMA_prior=False)
D = args.N_layers + 1
net = DUN_VI(model, depth_categorical_VI([1 / D] * D, cuda=True), X.shape[0], lr=args.lr, momentum=0.9, cuda=True,
             schedule=None, regression=True, pred_sig=None, weight_decay=args.wd)
out = train_fc_DUN(net, 'DUN_' + args.dataset, args.savedir, 2048, args.n_epochs, train, val, True, seed=0, nb_its_dev=5,
                   save_freq=5)
print('final ELBO estimate %.4f  train err %.4f' % (out[0][-1], out[3][-1]))
