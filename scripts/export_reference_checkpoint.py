#!/usr/bin/env python
"""Convert a checkpoint written by the reference's BaseNet.save (whole pickled objects, src/utils.py:124-154) into the
tensors-only layout of dun_amd.checkpoint, which dun_amd's wrappers read with `net.load_state(path)`.

    python scripts/export_reference_checkpoint.py /path/to/DUN-checkout theta_best.dat theta_best.state.pt

Run it ONCE, in the environment that wrote the checkpoint and only on files you wrote yourself: reading a pickled-object
checkpoint executes the class code it names (that is why the converted form exists).  The output holds state dicts only.
"""
import os
import sys

import torch


class _Holder(object):
    pass


def main(argv):
    if len(argv) != 4:
        raise SystemExit(__doc__)
    ref_root, src, dst = argv[1:]
    sys.path.insert(0, ref_root)  # the pickles name src.DUN...., src.probability....
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from dun_amd.checkpoint import net_state
    try:
        state = torch.load(src, map_location="cpu", weights_only=False)
    except TypeError:  # torch < 1.13 has no weights_only argument
        state = torch.load(src, map_location="cpu")
    net = _Holder()
    net.epoch, net.lr = state["epoch"], state["lr"]
    net.model, net.f_neg_loglike = state["model"], state["f_neg_loglike"]
    net.prob_model = state.get("prob_model")
    if net.prob_model is None:
        raise SystemExit("checkpoint has no prob_model (the reference's fallback branch, src/utils.py:137-143): the depth "
                         "posterior is part of a DUN's state; construct it and add it before exporting")
    net.optimizer, net.scheduler = state["optimizer"], state["scheduler"]
    torch.save(net_state(net, state.get("best_err")), dst)
    print("wrote", dst)


if __name__ == "__main__":
    main(sys.argv)
