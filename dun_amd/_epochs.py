"""Pieces the two epoch loops (train_fc.train_fc_DUN, train_img.train_loop) share."""
import numpy as np


def epoch_means(loader, step, width, flat_ims=False):
    """Row-weighted means of whatever ``step(x, y)`` returns (a ``width``-tuple of floats) over one pass."""
    sums, rows = np.zeros(width), 0
    for x, y in loader:
        xb = x.view(x.shape[0], -1) if flat_ims else x
        sums += np.asarray(step(xb, y), dtype=np.float64) * xb.shape[0]
        rows += len(xb)
    return sums / max(rows, 1)


class QWarmup(object):
    """q(d) warm-up: ``q_logits`` stays frozen for the first ``epochs`` epochs (src/DUN/train_fc.py:51-57,
    experiments/image/train_DUN.py:189-194).  Touches ``q_logits`` only when a warm-up was asked for, so wrappers
    whose probability model has none (DUN + depth_categorical) run through the same loops."""

    def __init__(self, prob_model, epochs):
        self.prob_model, self.epochs = prob_model, epochs
        if epochs > 0:
            prob_model.q_logits.requires_grad = False

    def at_epoch(self, epoch):
        if self.epochs > 0 and epoch == self.epochs:
            self.prob_model.q_logits.requires_grad = True
