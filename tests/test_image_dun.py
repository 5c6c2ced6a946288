"""Image DUNs (reduced C4/C5 family): constructor parity with the reference (same seed -> same weights, same
state_dict keys), and forward / ELBO / gradients / fit / eval / predict / fast_predict against golden vectors.
CPU variant drives the host logic with the stand-in backend; the GPU variant runs the HIP head + ELBO kernels with
the conv trunk on MIOpen."""
import contextlib
import io
import json
import os

os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")  # keep MIOpen's per-shape search short on a cold GPU box

import numpy as np
import pytest
import torch

from tests.util import Case, rel_err

CASES = ["img_r18", "img_r18_mnist"]
# the C4 / C5 constructors themselves: ResNet-50 DUN, start_depth=1, end_depth=13 (D = 12 exits), batch 4
CASES_R50 = ["img_r50_mnist", "img_r50_cifar"]


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def build(c, cuda):
    from dun_amd.probability import depth_categorical_VI
    from dun_amd.stochastic_img_resnets import resnet18, resnet50
    from dun_amd.training_wrappers import DUN_VI
    meta = json.loads(str(c["meta"]))
    torch.manual_seed(0)
    model = (resnet50 if "r50" in c.name else resnet18)(arch_uncert=True, **meta["kw"])
    D = meta["D"]
    if "channel_list" in meta:
        assert [int(v) for v in model.channel_list] == meta["channel_list"]
    assert model.n_layers == D
    pm = depth_categorical_VI([1.0 / D] * D, cuda=cuda)
    net = quiet(DUN_VI, model, pm, meta["N"], lr=meta["lr"], momentum=meta["momentum"], weight_decay=meta["wd"],
                cuda=cuda, regression=False)
    pm.q_logits.data.copy_(c.t("q_logits0"))
    torch.backends.cudnn.benchmark = False  # (the wrapper turns MIOpen's benchmark search on, as the reference does)
    return net, meta


def check(c, net, meta, dev, tol):
    model = net.model
    sd = model.state_dict()
    assert sorted(sd.keys()) == list(c["keys"])
    assert int(c["n_params"]) == sum(p.numel() for p in model.parameters())
    for k, v in c.sub("w/").items():
        assert np.array_equal(sd[k].detach().cpu().numpy().ravel()[:64], v), k  # same seed, same initial weights
    x, y = c.t("x").to(dev), c.t("y").long().to(dev)
    model.train()
    act = model.forward(x)
    assert rel_err(act.detach().cpu(), c["act_vec"]) < tol
    prior = model.get_w_prior_loglike()
    out = net.prob_model.fused_objective(prior, act, y, net.f_neg_loglike, meta["N"])
    assert rel_err(out.objective.cpu(), c["ELBO"]) < tol
    assert rel_err(out.dact.cpu(), c["g/act_vec"]) < 5 * tol
    net.optimizer.zero_grad()
    net._backward(act, prior, out)
    assert rel_err(model.output_block.weight.grad.cpu(), c["g/output_block.weight"]) < 5 * tol
    assert rel_err(model.output_block.bias.grad.cpu(), c["g/output_block.bias"]) < 5 * tol
    if c.has("g/conv1.0.weight"):
        assert rel_err(model.conv1[0].weight.grad.cpu(), c["g/conv1.0.weight"]) < 20 * tol
    assert rel_err(net.prob_model.q_logits.grad.cpu(), c["g/q_logits"]) < 5 * tol
    model.eval()
    with torch.no_grad():
        # (the train-mode forward above moved the BN running statistics, exactly as it did for the reference)
        assert rel_err(model.forward(x).cpu(), c["act_eval"]) < tol


@pytest.mark.parametrize("name", CASES)
def test_image_dun_cpu(name):
    from dun_amd import kernels
    from tests.cpu_backend import CpuBackend
    kernels.set_backend(CpuBackend())
    try:
        c = Case(name)
        net, meta = build(c, cuda=False)
        check(c, net, meta, "cpu", 2e-5)
        net2, _ = build(c, cuda=False)
        x, y = c.t("x"), c.t("y").long()
        for step in range(2):
            ret = net2.fit(x, y)
            assert np.allclose(ret, c["fit/ret"][step], rtol=1e-4 if step == 0 else 2e-3), (ret, c["fit/ret"][step])
        assert rel_err(net2.prob_model.q_logits.data, c["fit/q_logits"]) < 1e-3
        assert np.allclose(net2.eval(x, y), c["eval/ret"], rtol=5e-3)
        assert rel_err(net2.predict(x), c["pred"]) < 5e-3
        net2.prob_model.current_posterior = c.t("fast_pred_q")
        assert rel_err(quiet(net2.fast_predict, x), c["fast_pred"]) < 5e-3
    finally:
        kernels.set_backend(None)


@pytest.mark.parametrize("name", CASES_R50)
def test_resnet50_dun_constructor_and_step_cpu(name):
    """C4 / C5 model family under a test: keys, parameter count, seed-matched weights, adapt chaining up to end_depth=13,
    act_vec [12, B, 10], ELBO, head / q gradients and one fit step against the reference's numbers."""
    from dun_amd import kernels
    from tests.cpu_backend import CpuBackend
    kernels.set_backend(CpuBackend())
    try:
        c = Case(name)
        net, meta = build(c, cuda=False)
        assert meta["D"] == 12 and net.model.n_layers == 12 and net.model.output_block.in_features == 2048
        check(c, net, meta, "cpu", 5e-5)
        net2, _ = build(c, cuda=False)
        ret = net2.fit(c.t("x"), c.t("y").long())
        assert np.allclose(ret, c["fit/ret"][0], rtol=2e-4), (ret, c["fit/ret"][0])
    finally:
        kernels.set_backend(None)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES_R50)
def test_resnet50_dun_gpu(name):
    """The same fixtures through the HIP head + CE/ELBO kernels (conv trunk on MIOpen, hence the looser tolerance)."""
    assert torch.cuda.is_available()
    c = Case(name)
    net, meta = build(c, cuda=True)
    check(c, net, meta, "cuda", 5e-3)
    net2, _ = build(c, cuda=True)
    ret = net2.fit(c.t("x"), c.t("y").long())
    assert np.allclose(ret, c["fit/ret"][0], rtol=2e-2), (ret, c["fit/ret"][0])


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_image_dun_gpu(name):
    assert torch.cuda.is_available()
    c = Case(name)
    net, meta = build(c, cuda=True)
    check(c, net, meta, "cuda", 2e-3)  # conv trunk: MIOpen vs the reference's CPU convolutions
    net2, _ = build(c, cuda=True)
    x, y = c.t("x"), c.t("y").long()
    for step in range(2):
        ret = net2.fit(x, y)
        assert np.allclose(ret, c["fit/ret"][step], rtol=5e-3 if step == 0 else 5e-2), (ret, c["fit/ret"][step])
    assert np.allclose(net2.eval(x, y), c["eval/ret"], rtol=0.1)
    assert rel_err(net2.predict(x).cpu(), c["pred"]) < 0.1
    net2.prob_model.current_posterior = c.t("fast_pred_q").cuda()
    assert rel_err(quiet(net2.fast_predict, x).cpu(), c["fast_pred"]) < 0.1


def _epoch_loop_case(tmp_path, cuda):
    """train_img.train_loop (reference: experiments/image/train_DUN.py:145-248) on a tiny synthetic MNIST-shaped set."""
    import pickle
    from dun_amd.train_img import make_image_dun, train_loop
    torch.manual_seed(0)
    net = quiet(make_image_dun, dataset="MNIST", model="resnet18", start_depth=1, end_depth=5, lr=0.05,
                milestones=[2], cuda=cuda)
    torch.backends.cudnn.benchmark = False
    g = torch.Generator().manual_seed(1)
    x = torch.randn(96, 1, 28, 28, generator=g)
    y = (x.mean(dim=(1, 2, 3)) > 0).long() + 2 * (x[:, 0, :14].mean(dim=(1, 2)) > 0).long()   # learnable 4-class labels
    batches = [(x[i:i + 32], y[i:i + 32]) for i in range(0, 96, 32)]
    q0 = net.prob_model.q_logits.data.clone()
    out = quiet(train_loop, net, batches, batches[:1], 96, epochs=3, savedir=str(tmp_path), q_nograd_its=1)
    elbo, err_tr, nll_tr, err_dev, nll_dev, best = out
    assert np.all(np.isfinite(elbo)) and np.all(np.isfinite(nll_tr)) and np.all(np.isfinite(nll_dev))
    assert elbo[-1] > elbo[0] and nll_tr[-1] < nll_tr[0]
    assert net.epoch == 3 and abs(net.optimizer.param_groups[0]["lr"] - 0.005) < 1e-12   # milestone at epoch 2
    assert net.prob_model.q_logits.requires_grad and not torch.equal(net.prob_model.q_logits.data.cpu(), q0.cpu())
    assert best == min(1.0, err_dev.min())
    for f in ("checkpoint.pth.tar", "stats_array.pkl", "model_best.pth.tar"):
        assert os.path.exists(os.path.join(str(tmp_path), f)), f
    with open(os.path.join(str(tmp_path), "stats_array.pkl"), "rb") as f:   # written by the loop just above
        stored = pickle.load(f)
    # on-disk order of the reference: marginal_loglike, err_train, train_loss, err_dev, dev_loss
    assert len(stored) == 5 and all(np.array_equal(a, b) for a, b in zip(stored, out[:5]))
    # resume: two more epochs continue the arrays from the checkpointed epoch
    net2 = quiet(make_image_dun, dataset="MNIST", model="resnet18", start_depth=1, end_depth=5, lr=0.05,
                 milestones=[2], cuda=cuda)
    out2 = quiet(train_loop, net2, batches, batches[:1], 96, epochs=5, savedir=str(tmp_path),
                 resume=os.path.join(str(tmp_path), "checkpoint.pth.tar"))
    assert net2.epoch == 5 and np.allclose(out2[0][:3], elbo) and np.all(out2[0][3:] != 0)
    # every epoch kept, and a resume path that does not exist starts from scratch
    net3 = quiet(make_image_dun, dataset="MNIST", model="resnet18", start_depth=1, end_depth=5, lr=0.05, cuda=cuda)
    sub = os.path.join(str(tmp_path), "all")
    out3 = quiet(train_loop, net3, batches[:1], batches[:1], 32, epochs=2, savedir=sub, save_all_epochs=True,
                 resume=os.path.join(sub, "nothing_here.pth.tar"))
    assert os.path.exists(os.path.join(sub, "0_checkpoint.pth.tar")) and os.path.exists(os.path.join(sub, "1_checkpoint.pth.tar"))
    assert net3.epoch == 2 and np.all(out3[0] != 0)


def test_image_epoch_loop_cpu(tmp_path):
    from dun_amd import kernels
    from tests.cpu_backend import CpuBackend
    kernels.set_backend(CpuBackend())
    try:
        _epoch_loop_case(tmp_path, cuda=False)
    finally:
        kernels.set_backend(None)


@pytest.mark.gpu
def test_image_epoch_loop_trains_checkpoints_and_resumes(tmp_path):
    _epoch_loop_case(tmp_path, cuda=True)
