"""Data feeding for the FC configs: the toy 'wiggle' regression set regenerated from its closed
form (reference: src/datasets/additional_gap_loader.py:226-250), the UCI "gap"/standard split reader over a directory
in the reference's on-disk layout (src/datasets/UCI_gap_loader.py:12-60; no download: there is no network), readers
for the three toy sets the reference keeps as .npy files (agw_1d, matern_1d, my_1d: additional_gap_loader.py:65-223;
parity unpinned: the reference has no fixture for them and its data files are not copied here -- tested on synthetic
files of the same layout), and a
device-resident batch iterator that keeps the epoch loop free of DataLoader workers and host-to-device copies.
Loaders that download or read image corpora are out of scope."""
import os

import numpy as np
import torch
import torch.utils.data as data


def load_wiggle():
    rs = np.random.RandomState(0)
    n = 300
    x = rs.randn(n) * 2.5 + 5
    y = np.sin(np.pi * x) + 0.2 * np.cos(np.pi * x * 4) - 0.3 * x + rs.randn(n) * 0.25
    x, y = x[:, None], y[:, None]
    X = ((x - x.mean(axis=0)) / x.std(axis=0)).astype(np.float32)
    Y = ((y - y.mean(axis=0)) / y.std(axis=0)).astype(np.float32)
    return X, Y


def _toy_file(base_dir, rel):
    path = os.path.join(base_dir, rel)
    if not os.path.exists(path):
        raise FileNotFoundError("%s not found (the reference downloads or regenerates it; point base_dir at a DUN "
                                "checkout's experiments/data)" % path)
    return np.load(path, allow_pickle=False)


def load_agw_1d(base_dir, get_feats=False):
    """The 1-d regression set of Izmailov et al. as the reference reads it (src/datasets/additional_gap_loader.py:65-90,
    file <base_dir>/agw_data/data.npy, columns x, y): standardised inputs [n,1] (or, get_feats, the standardised
    quadratic features [x/2, (x/2)^2], [n,2]) and standardised targets [n,1], float32.  No download."""
    d = _toy_file(base_dir, os.path.join("agw_data", "data.npy"))
    x, y = d[:, 0], d[:, 1][:, None]
    f = np.hstack([x[:, None] / 2.0, (x[:, None] / 2.0) ** 2])
    X = ((x - x.mean(axis=0)) / x.std(axis=0)).astype(np.float32)
    Y = ((y - y.mean(axis=0)) / y.std(axis=0)).astype(np.float32)
    F = ((f - f.mean(axis=0)) / f.std(axis=0)).astype(np.float32)
    return (F, Y) if get_feats else (X[:, None], Y)


def load_matern_1d(base_dir):
    """Stored draw of the two-cluster Matern GP set (additional_gap_loader.py:113-158, file
    <base_dir>/matern_data/matern_1d.npy, columns x, y): x [n,1], y [n,1] as stored.  The reference regenerates the draw
    with GPy when the file is missing; here a missing file raises."""
    xy = _toy_file(base_dir, os.path.join("matern_data", "matern_1d.npy"))
    return xy[:, 0][:, None], xy[:, 1][:, None]


def load_my_1d(base_dir):
    """Stored draw of the reference's own 1-d set (additional_gap_loader.py:161-223, file
    <base_dir>/my_1d_data/my_1d_data.npy, columns X_train, y_train (homoscedastic), X_test, y_test): each [n,1] float32."""
    xy = _toy_file(base_dir, os.path.join("my_1d_data", "my_1d_data.npy"))
    return tuple(xy[:, i, None].astype(np.float32) for i in range(4))


class Datafeed(data.Dataset):
    """Tensor dataset with optional targets/transform (as src/utils.py:202-219)."""

    def __init__(self, x_train, y_train=None, transform=None):
        self.data = x_train
        self.targets = y_train
        self.transform = transform

    def __getitem__(self, index):
        img = self.data[index]
        if self.transform is not None:
            img = self.transform(img)
        if self.targets is not None:
            return img, self.targets[index]
        return img

    def __len__(self):
        return len(self.data)


def load_gap_UCI(base_dir, dname, n_split=0, gap=True):
    """Same returns as the reference's load_gap_UCI: (X_train, X_test, x_means, x_stds, y_train, y_test, y_means,
    y_stds), features/targets standardised with the TRAIN split's statistics (stds under 1e-10 -> 1).  Reads
    <base_dir>/UCI_for_sharing/{standard,gap}/<dname>/data/{data,index_features,index_target,index_train_i,
    index_test_i}.txt; a missing directory raises instead of downloading."""
    root = os.path.join(base_dir, "UCI_for_sharing")
    if not os.path.exists(root):
        raise FileNotFoundError("%s not found (the reference fetches it from the network; place the extracted "
                                "UCI_for_sharing directory there)" % root)
    dir_load = os.path.join(root, "standard", dname, "data")
    dir_idx = os.path.join(root, "gap" if gap else "standard", dname, "data")
    table = np.loadtxt(os.path.join(dir_load, "data.txt"))
    feature_idx = np.loadtxt(os.path.join(dir_load, "index_features.txt")).astype(int)
    target_idx = np.loadtxt(os.path.join(dir_load, "index_target.txt")).astype(int)
    splits = []
    for i in range(20):
        tr, te = (os.path.join(dir_idx, "index_%s_%d.txt" % (k, i)) for k in ("train", "test"))
        if os.path.exists(tr) and os.path.exists(te):
            splits.append((np.loadtxt(tr).astype(int), np.loadtxt(te).astype(int)))
    train_idx, test_idx = splits[n_split]
    rows_train, rows_test = table[train_idx, :], table[test_idx, :]
    X_train = rows_train[:, feature_idx].astype(np.float32)
    X_test = rows_test[:, feature_idx].astype(np.float32)
    y_train = rows_train[:, target_idx].astype(np.float32)
    y_test = rows_test[:, target_idx].astype(np.float32)
    x_means, x_stds = X_train.mean(axis=0), X_train.std(axis=0)
    y_means, y_stds = y_train.mean(axis=0), y_train.std(axis=0)
    x_stds[x_stds < 1e-10] = 1.
    X_train = (X_train - x_means) / x_stds
    X_test = (X_test - x_means) / x_stds
    y_train = ((y_train - y_means) / y_stds)[:, np.newaxis]
    y_test = ((y_test - y_means) / y_stds)[:, np.newaxis]
    return X_train, X_test, x_means, x_stds, y_train, y_test, y_means, y_stds


class DeviceLoader(object):
    """Batches of a dataset that lives on the device: `for x, y in loader` yields slices (views when not shuffling,
    one gather per batch when shuffling), `len(loader.dataset)` is the row count the sweeps ask for.  Stands where
    train_fc_DUN takes a torch DataLoader; `rank`/`world` hand every rank its rows of each global batch."""

    def __init__(self, x, y, batch_size, shuffle=False, device=None, seed=0, drop_last=False, rank=0, world=1):
        device = device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu")
        self.x = torch.as_tensor(x).to(device)
        self.y = torch.as_tensor(y).to(device)
        assert self.x.shape[0] == self.y.shape[0]
        assert batch_size % world == 0, "global batch must split evenly over the ranks"
        self.batch_size, self.shuffle, self.drop_last = batch_size, shuffle, drop_last
        self.rank, self.world = rank, world
        self.dataset = Datafeed(self.x, self.y)
        self._gen = torch.Generator(device="cpu").manual_seed(seed)  # same permutation on every rank

    def __len__(self):
        n = self.x.shape[0]
        return n // self.batch_size if self.drop_last or self.world > 1 else (n + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        n = self.x.shape[0]
        perm = torch.randperm(n, generator=self._gen).to(self.x.device) if self.shuffle else None
        local = self.batch_size // self.world
        for i in range(len(self)):
            lo = i * self.batch_size + self.rank * local
            hi = min(lo + local, n) if self.world > 1 else min((i + 1) * self.batch_size, n)
            if perm is None:
                yield self.x[lo:hi], self.y[lo:hi]
            else:
                idx = perm[lo:hi]
                yield self.x[idx], self.y[idx]
