"""The one synthetic dataset the plumbing config needs (BASELINE.json configs[0]): the toy 'wiggle' regression
set, regenerated from its closed form (reference: src/datasets/additional_gap_loader.py:226-250).  Loaders that
download or read image corpora are out of scope."""
import numpy as np
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
