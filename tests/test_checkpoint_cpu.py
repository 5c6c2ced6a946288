"""Tensors-only checkpoints (dun_amd.checkpoint) on the CPU stand-in backend: a state written from
the REFERENCE's own objects (tests/golden/ckpt_ref_vi.pt, made by make_golden.checkpoint_case) is ingested -- weights,
BatchNorm buffers, q, log_std, per-parameter momentum, schedule -- and the continued training equals the reference's;
save_state -> load_state round-trips this package's wrappers."""
import contextlib
import io
import os

import numpy as np
import pytest
import torch

from tests.util import GOLDEN, Case, rel_err


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


@pytest.fixture
def cpu_backend():
    from dun_amd import kernels
    from tests.cpu_backend import CpuBackend
    kernels.set_backend(CpuBackend())
    yield
    kernels.set_backend(None)


def build(cfg, seed, schedule=None, vi=True, cuda=False):
    from dun_amd.probability import depth_categorical, depth_categorical_VI
    from dun_amd.stochastic_fc_models import arq_uncert_fc_resnet
    from dun_amd.training_wrappers import DUN, DUN_VI
    torch.manual_seed(seed)
    D = cfg["L"] + 1
    model = arq_uncert_fc_resnet(cfg["din"], cfg["O"], cfg["W"], cfg["L"])
    pm = (depth_categorical_VI if vi else depth_categorical)([1.0 / D] * D, cuda=cuda)
    # deliberately different hyper-parameters: the checkpoint's optimizer state must win
    return quiet(DUN_VI if vi else DUN, model, pm, cfg["N"], lr=1e-3, momentum=0.1, weight_decay=0.0, cuda=cuda,
                 regression=True, schedule=schedule)


def continue_like_reference(net, c, dev="cpu"):
    x, y = c.t("x").to(dev), c.t("y").to(dev)
    rets = [net.fit(x, y)]
    net.update_lr()
    assert abs(net.optimizer.param_groups[0]["lr"] - float(c["lr_after"])) < 1e-12
    rets.append(net.fit(x, y))
    assert np.allclose(np.array(rets), c["fit/ret"], rtol=2e-4, atol=1e-6), (rets, c["fit/ret"])
    sd = net.model.state_dict()
    for k, v in c.sub("fit/sd/").items():
        if "num_batches_tracked" in k:
            assert int(sd[k]) == int(v), k
        else:
            assert rel_err(sd[k].cpu(), v) < 2e-4, k
    assert rel_err(net.prob_model.q_logits.data.cpu(), c["fit/q_logits"]) < 2e-4
    assert rel_err(net.f_neg_loglike.log_std.data.cpu(), c["fit/log_std"]) < 2e-4


@pytest.mark.parametrize("schedule", [None, [2, 4]])
def test_reference_state_is_ingested_and_training_continues_like_the_reference(cpu_backend, schedule):
    c = Case("ckpt_ref_vi")
    net = build(c.cfg, seed=99, schedule=schedule)
    epoch, best = quiet(net.load_state, os.path.join(GOLDEN, "ckpt_ref_vi.pt"))
    assert (epoch, best) == (1, 0.25) and net.epoch == 1 and net.lr == c.cfg["lr"]
    g = net.optimizer.param_groups[0]
    assert (g["lr"], g["momentum"], g["weight_decay"]) == (c.cfg["lr"], c.cfg["momentum"], c.cfg["wd"])
    continue_like_reference(net, c)


def test_fixture_holds_data_only():
    st = torch.load(os.path.join(GOLDEN, "ckpt_ref_vi.pt"), weights_only=True)  # would raise on any pickled object
    assert st["format"] == "dun.state/1" and len(st["optimizer"]["state"]) == 18
    assert st["scheduler"]["milestones"] == [2, 4]


def test_round_trip(cpu_backend, tmp_path):
    c = Case("ckpt_ref_vi")
    x, y = c.t("x"), c.t("y")
    a = build(c.cfg, seed=3, schedule=[1, 3])
    a.optimizer.param_groups[0].update(lr=2e-2, momentum=0.8, weight_decay=1e-3)
    for _ in range(2):
        a.fit(x, y)
    a.update_lr()
    path = str(tmp_path / "state.pt")
    quiet(a.save_state, path)
    torch.load(path, weights_only=True)
    b = build(c.cfg, seed=4)
    assert quiet(b.load_state, path) == 1
    for _ in range(2):
        ra, rb = a.fit(x, y), b.fit(x, y)
        assert np.allclose(ra, rb, rtol=1e-6, atol=1e-7), (ra, rb)
        a.update_lr()
        b.update_lr()
    assert a.optimizer.param_groups[0]["lr"] == b.optimizer.param_groups[0]["lr"]
    assert torch.allclose(a.model.flat_params, b.model.flat_params, rtol=1e-6, atol=1e-8)
    assert torch.equal(a.model.num_batches_tracked, b.model.num_batches_tracked)
    assert torch.allclose(a.prob_model.current_posterior, b.prob_model.current_posterior)


def test_state_before_the_first_step_and_rejections(cpu_backend, tmp_path):
    c = Case("ckpt_ref_vi")
    a = build(c.cfg, seed=5)
    path = str(tmp_path / "fresh.pt")
    quiet(a.save_state, path)
    st = torch.load(path, weights_only=True)
    assert st["optimizer"]["state"] == {} and st["scheduler"] is None
    b = build(c.cfg, seed=6)
    b.fit(c.t("x"), c.t("y"))  # b has momentum; the checkpoint has none: it must be cleared
    quiet(b.load_state, path)
    x, y = c.t("x"), c.t("y")
    assert np.allclose(a.fit(x, y), b.fit(x, y), rtol=1e-6)
    assert torch.allclose(a.model.flat_params, b.model.flat_params, rtol=1e-6, atol=1e-8)
    with pytest.raises(ValueError, match="format"):
        b.load_state({"format": "something else"})
    mll = build(c.cfg, seed=7, vi=False)
    with pytest.raises(ValueError, match="q_logits"):
        quiet(mll.load_state, os.path.join(GOLDEN, "ckpt_ref_vi.pt"))
    small = build(dict(c.cfg, L=2), seed=8)
    before = small.model.flat_params.clone()
    with pytest.raises(ValueError, match="parameters"):
        quiet(small.load_state, os.path.join(GOLDEN, "ckpt_ref_vi.pt"))
    assert torch.equal(before, small.model.flat_params)  # refused before anything was overwritten
    # same parameter COUNT, another width: also refused before the first byte is overwritten (hyper-parameters included)
    wide = build(dict(c.cfg, W=c.cfg["W"] + 4), seed=9)
    before = wide.model.flat_params.clone()
    lr_before = wide.optimizer.param_groups[0]["lr"]
    q_before = wide.prob_model.q_logits.detach().clone()
    with pytest.raises(ValueError, match="shape"):
        quiet(wide.load_state, os.path.join(GOLDEN, "ckpt_ref_vi.pt"))
    assert torch.equal(before, wide.model.flat_params) and torch.equal(q_before, wide.prob_model.q_logits.detach())
    assert wide.optimizer.param_groups[0]["lr"] == lr_before

