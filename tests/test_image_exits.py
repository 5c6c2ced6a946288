"""Everything behind the convolutional trunk of the C5 image DUN, against the reference at 1e-5.

The trunk's convolutions run on MIOpen, whose algorithms differ from the reference's CPU convolutions at the 1e-3
level (tests/test_image_dun.py) -- a tolerance behind which an error in the parts this repository owns could hide.
Here the trunk is taken out of the comparison: tests/golden/img_r50_cifar_exits.npz holds the pooled [D,B,2048]
features that reached the reference's shared output_block at each of its 12 exits (captured by a hook, see
tests/golden/make_golden.py image_exits_case), and the test feeds them in where `_exit_features` would produce
them.  From there on it is the product path: `_SharedHead` (HIP head kernels, forward and backward), the fused
softmax-CE / ELBO objective and its gradients, `eval`, `predict`, and `fast_forward_impl` / `fast_predict`'s exit
masking (reference: src/DUN/stochastic_img_resnets.py:288-315, src/DUN/training_wrappers.py:95-169).
"""
import contextlib
import io
import json

import numpy as np
import pytest
import torch

from tests.util import Case, rel_err

TOL = 1e-5
ZERO_ROWS = [0, 1, 2, 3, 4, 6, 8, 9, 10]   # posterior mass 0 (eight exits) or 5e-3 (exit 2): not above min_prob = 1e-2


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def build(c, cuda):
    from dun_amd.probability import depth_categorical_VI
    from dun_amd.stochastic_img_resnets import resnet50
    from dun_amd.training_wrappers import DUN_VI
    meta = json.loads(str(c["meta"]))
    torch.manual_seed(0)
    model = resnet50(arch_uncert=True, **meta["kw"])
    D = meta["D"]
    assert model.n_layers == D == 12
    with torch.no_grad():
        model.output_block.weight.copy_(c.t("output_block.weight"))
        model.output_block.bias.copy_(c.t("output_block.bias"))
    pm = depth_categorical_VI([1.0 / D] * D, cuda=cuda)
    net = quiet(DUN_VI, model, pm, meta["N"], lr=meta["lr"], momentum=meta["momentum"], weight_decay=meta["wd"],
                cuda=cuda, regression=False)
    pm.q_logits.data.copy_(c.t("q_logits0"))
    torch.backends.cudnn.benchmark = False
    return net, meta


def stub_trunk(model, feats):
    """Blocks become identities and `_exit_features` hands out the reference's pooled features of that exit; the
    loops over exits, the stacking, the shared head and the masking stay the product's own."""
    def ident(x):
        return x
    model.conv1.forward = ident
    for m in model.layer_list:
        m.forward = ident
    model._exit_features = lambda x, layer_idx: feats[layer_idx - model.start_depth]


def run(dev):
    c = Case("img_r50_cifar_exits")
    cuda = dev == "cuda"
    net, meta = build(c, cuda)
    model, pm = net.model, net.prob_model
    x, y = c.t("x").to(dev), c.t("y").long().to(dev)

    # ---- train mode: head forward, objective, and the gradients the head hands back
    feats = c.t("train/feat").to(dev).requires_grad_(True)
    stub_trunk(model, feats)
    model.train()
    act = model.forward(x)
    assert act.shape == (12, 4, 10)
    assert rel_err(act.detach().cpu(), c["train/act_vec"]) < TOL
    prior = model.get_w_prior_loglike()
    out = pm.fused_objective(prior, act, y, net.f_neg_loglike, meta["N"])
    assert rel_err(out.objective.cpu(), c["train/ELBO"]) < TOL
    assert rel_err(out.dact.cpu(), c["train/g_act_vec"]) < TOL
    net.optimizer.zero_grad()
    net._backward(act, prior, out)
    assert rel_err(model.output_block.weight.grad.cpu(), c["train/g_output_block.weight"]) < TOL
    assert rel_err(model.output_block.bias.grad.cpu(), c["train/g_output_block.bias"]) < TOL
    assert rel_err(pm.q_logits.grad.cpu(), c["train/g_q_logits"]) < TOL
    assert rel_err(feats.grad.cpu(), c["train/g_feat"]) < TOL
    # the autograd route of the reference's own call sequence lands on the same numbers
    act2 = model.forward(x)
    ELBO = pm.estimate_ELBO(prior, act2, y, net.f_neg_loglike, meta["N"], Beta=1)
    assert rel_err(ELBO.detach().cpu(), c["train/ELBO"]) < TOL

    # ---- eval mode: act_vec, eval, predict
    with torch.no_grad():
        stub_trunk(model, c.t("eval/feat").to(dev))
        model.eval()
        assert rel_err(model.forward(x).cpu(), c["eval/act_vec"]) < TOL
        pm.current_posterior = c.t("posterior").to(dev)
        assert np.allclose(np.array(net.eval(x, y), dtype=np.float64), c["eval/ret"], rtol=TOL, atol=0)
        pred = net.predict(x)
        assert rel_err(pred.cpu(), c["pred"]) < TOL
        assert np.allclose(pred.sum(1).cpu().numpy(), 1.0, atol=1e-6)

        # ---- early-exit inference: which exits are evaluated, and what comes out
        q = c.t("fast/q").to(dev)
        assert float(q[2]) == pytest.approx(5e-3) and float(q[5]) == pytest.approx(2e-2)
        fa = quiet(model.fast_forward_impl, x, q, min_prob=1e-2).cpu().numpy()
        zero = [d for d in range(12) if not fa[d].any()]
        assert zero == ZERO_ROWS, zero                                    # exit 2 (5e-3) skipped, exit 5 (2e-2) kept
        assert [d for d in range(12) if not c["fast/act_vec"][d].any()] == ZERO_ROWS
        assert rel_err(fa, c["fast/act_vec"]) < TOL
        for d in ZERO_ROWS:
            assert np.array_equal(fa[d], np.zeros_like(fa[d]))            # exactly zero, not merely small
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            model.fast_forward_impl(x, q, min_prob=1e-2)
        skipped = [int(line.split()[-1]) for line in buf.getvalue().splitlines() if line.startswith("skipping layer")]
        assert skipped == [d + model.start_depth for d in ZERO_ROWS]      # the reference's message, per skipped block index
        # the synchronisation-free form: all exits evaluated, low-mass rows zeroed on the device
        fd = model.fast_forward_impl(x, q, min_prob=1e-2, device_mask=True).cpu().numpy()
        assert [d for d in range(12) if not fd[d].any()] == ZERO_ROWS and rel_err(fd, fa) < 1e-6
        for d in ZERO_ROWS:
            assert np.array_equal(fd[d], np.zeros_like(fd[d]))
        pm.current_posterior = q
        assert rel_err(quiet(net.fast_predict, x).cpu(), c["fast/pred"]) < TOL
        # min_prob is a strict threshold (layer_probs > min_prob): mass exactly at it is skipped
        q2 = q.clone()
        q2[5] = 1e-2
        fa2 = quiet(model.fast_forward_impl, x, q2, min_prob=1e-2).cpu().numpy()
        assert not fa2[5].any() and fa2[7].any()
        # nothing above the threshold: all rows zero, no head launch on an empty stack
        fa3 = quiet(model.fast_forward_impl, x, torch.full((12,), 1e-3, device=dev), min_prob=1e-2)
        assert fa3.shape == (12, 4, 10) and not bool(fa3.any())


def test_image_exits_cpu():
    from dun_amd import kernels
    from tests.cpu_backend import CpuBackend
    kernels.set_backend(CpuBackend())
    try:
        run("cpu")
    finally:
        kernels.set_backend(None)


@pytest.mark.gpu
def test_image_exits_gpu():
    assert torch.cuda.is_available()
    run("cuda")
