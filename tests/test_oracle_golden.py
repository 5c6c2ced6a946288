"""Pin the CPU oracle against vectors produced by the real reference (tests/golden/make_golden.py)."""
import numpy as np
import pytest
import torch

from oracle import dun_oracle as O
from tests.util import Case, FC_CASES, PRIOR_CASES, rel_err, stacked_grads_from_reference

TOL = 2e-5  # same ATen CPU ops on both sides; thread-count dependent summation order leaves ~5e-6


@pytest.fixture(autouse=True)
def _four_threads():
    """The goldens were generated with 4 intra-op threads; with the same count the oracle reproduces the
    reference bit for bit on the build container, other counts move fp32 results at the 1e-6..1e-3 level
    (gradients through ten BatchNorm layers are that ill-conditioned in fp32)."""
    n = torch.get_num_threads()
    torch.set_num_threads(4)
    yield
    torch.set_num_threads(n)


def _noise_bounded(got32, ref32, truth64, slack=4.0, floor=2e-6):
    """fp32 parity in the presence of ill-conditioning: our error against the fp64 truth must not exceed a
    small multiple of the reference's own fp32 error against the same truth."""
    return rel_err(got32, truth64) <= slack * rel_err(ref32, truth64) + floor


@pytest.mark.parametrize("name", FC_CASES + PRIOR_CASES)
def test_forward_and_probability_chain(name):
    c = Case(name)
    cfg = c.cfg
    net = c.oracle_net()
    x, y = c.t("x"), c.y()
    p = O.clone_params(net.p)
    act = O.fc_forward(p, x, train=True, residual=not cfg.get("mlp", False))
    assert rel_err(act, c["act_vec"]) < TOL
    ll = O.loglike_per_act(act, y, cfg["regression"], net.log_std)
    assert rel_err(ll, c["loglike"]) < TOL
    prior = net.prior_loglikes()
    if c.has("prior_ll"):  # zeros [D] without a weight prior, a scalar (summed over layers) with one
        assert prior.shape == c["prior_ll"].shape and rel_err(prior, c["prior_ll"]) < TOL
    joint = O.joint_loglike(prior, act, y, cfg["regression"], cfg["N"], net.log_std)
    assert rel_err(joint, c["joint"]) < TOL
    lm = O.marg_loglike(joint, net.logprior)
    assert rel_err(lm, c["log_marg"]) < TOL
    assert np.allclose(O.depth_log_posterior(joint, net.logprior, lm).numpy(), c["log_post"], rtol=1e-5, atol=1e-4)
    if cfg["vi"]:
        assert rel_err(O.q_probs(net.q_logits), c["q"]) < TOL
        assert abs(float(O.kl_q_prior(net.q_logits, net.logprior)) - float(c["KL"])) < 1e-6
        e = O.elbo(prior, act, y, cfg["regression"], cfg["N"], net.q_logits, net.logprior, net.log_std)
        assert rel_err(e, c["ELBO"]) < TOL
    # BN running statistics after one train-mode forward
    after = c.sub("sd_after_fwd/")
    for i in range(cfg["L"]):
        assert rel_err(p["running_mean"][i], after["stochstic_layers.%d.block.2.running_mean" % i]) < TOL
        assert rel_err(p["running_var"][i], after["stochstic_layers.%d.block.2.running_var" % i]) < TOL
        assert int(p["num_batches_tracked"][i]) == int(after["stochstic_layers.%d.block.2.num_batches_tracked" % i])


@pytest.mark.parametrize("name", FC_CASES + PRIOR_CASES)
def test_gradients_and_closed_form(name):
    c = Case(name)
    cfg = c.cfg
    net = c.oracle_net()
    x, y = c.t("x"), c.y()
    ret, grads, act, joint = net.fit(x, y, return_grads=True)
    net64 = c.oracle_net(torch.float64)
    _, g64, _, _ = net64.fit(x.double(), y if y.dtype == torch.int64 else y.double(), return_grads=True)
    ref = stacked_grads_from_reference(c)
    for k, v in ref.items():
        assert _noise_bounded(grads[k], v, g64[k]), k
    if cfg["vi"]:
        assert _noise_bounded(grads["q_logits"], c["g/q_logits"], g64["q_logits"])
    if cfg["regression"] and cfg.get("pred_sig") is None:
        assert _noise_bounded(grads["log_std"], c["g/log_std"], g64["log_std"])
    # hand-derived backward of the probability module (what the HIP kernel implements)
    net0 = c.oracle_net()
    w = O.q_probs(net0.q_logits) if cfg["vi"] else O.depth_log_posterior(joint, net0.logprior).exp()
    cf = O.elbo_grads_closed_form(act, y, cfg["regression"], cfg["N"], w, log_std=net0.log_std, vi=cfg["vi"],
                                  q_logits=net0.q_logits if cfg["vi"] else None, logprior=net0.logprior,
                                  prior_loglikes=net0.prior_loglikes())
    assert rel_err(cf["dact"], c["g/act_vec"]) < 2e-5
    if cfg["vi"]:
        assert rel_err(cf["dq_logits"], c["g/q_logits"]) < 2e-4
    if cfg["regression"] and cfg.get("pred_sig") is None:
        assert rel_err(cf["dlog_std"], c["g/log_std"]) < 2e-5


@pytest.mark.parametrize("name", FC_CASES + PRIOR_CASES)
def test_fit_trajectory_and_inference(name):
    c = Case(name)
    cfg = c.cfg
    net = c.oracle_net()
    x, y = c.t("x"), c.y()
    for step in range(3):
        ret = net.fit(x, y)
        # step 0 is a pure function of the inputs; later steps amplify fp32 noise through the update
        rt = 2e-5 if step == 0 else 2e-4
        assert np.allclose(ret, c["fit/ret"][step], rtol=rt, atol=1e-6), (step, ret, c["fit/ret"][step])
        assert np.allclose(net.current_posterior.numpy(), c["fit/posterior"][step], rtol=1e-4, atol=1e-7)
    sd = c.sub("fit/sd/")
    L = cfg["L"]
    assert rel_err(net.p["w_in"], sd["input_layer.weight"]) < 1e-4
    assert rel_err(net.p["w_out"], sd["output_layer.weight"]) < 1e-4
    for i in range(L):
        assert rel_err(net.p["w"][i], sd["stochstic_layers.%d.block.0.weight" % i]) < 1e-4
        assert rel_err(net.p["gamma"][i], sd["stochstic_layers.%d.block.2.weight" % i]) < 1e-4
        assert rel_err(net.p["running_var"][i], sd["stochstic_layers.%d.block.2.running_var" % i]) < 1e-4
    if cfg["vi"]:
        assert rel_err(net.q_logits, c["fit/q_logits"]) < 1e-4
    if cfg["regression"]:
        assert rel_err(net.log_std, c["fit/log_std"]) < 1e-4
    assert np.allclose(net.evaluate(x, y), c["eval/ret"], rtol=2e-4)
    xt = c.t("x_test")
    if cfg["regression"]:
        mu, std = net.predict(xt, get_std=True)
        assert rel_err(mu, c["pred/mu"]) < 2e-4 and rel_err(std, c["pred/std"]) < 2e-4
        _, mstd = net.predict(xt, get_std=True, return_model_std=True)
        assert np.allclose(mstd.numpy(), c["pred/model_std"], rtol=1e-3, atol=1e-5, equal_nan=True)
        assert rel_err(net.predict(xt), c["pred/plain"]) < 2e-4
        if c.has("pred/depth_k"):
            k = int(c["pred/depth_k"])
            net.current_posterior = c.t("pred/depth_k_q")
            mu, _ = net.predict(xt, depth=k, get_std=True, return_model_std=True)
            assert rel_err(mu, c["pred/depth_k_mu"]) < 2e-4
    else:
        assert rel_err(net.predict(xt), c["pred/plain"]) < 2e-4
    if cfg.get("sweep"):
        bs = int(c["sweep/batch"])
        N = cfg["N"]
        net.current_posterior = torch.from_numpy(c["fit/posterior"][2])
        batches = [(x[i:i + bs], y[i:i + bs]) for i in range(0, N, bs)]
        post, lm = net.exact_d_posterior(batches)
        assert np.allclose(post.numpy(), c["sweep/posterior"], rtol=1e-4, atol=1e-7)
        assert abs(lm - float(c["sweep/log_marg"])) < 2e-5 * abs(lm)
        if cfg["vi"]:
            assert abs(net.exact_elbo(batches) - float(c["sweep/ELBO"])) < 2e-5 * abs(float(c["sweep/ELBO"]))


def test_edge_cases_regression():
    e = Case("edge")
    for tag in ("clamp", "peaked", "one_depth", "one_row"):
        p = "reg_%s/" % tag
        act, y = e.t(p + "act"), e.t(p + "y")
        ql, ls = e.t(p + "q_logits"), e.t(p + "log_std")
        N = int(e[p + "N"])
        D = act.shape[0]
        logprior = torch.full((D,), 1.0 / D).log()
        joint = O.joint_loglike(torch.zeros(D), act, y, True, N, ls)
        assert rel_err(joint, e[p + "joint"]) < TOL
        assert rel_err(O.elbo(torch.zeros(D), act, y, True, N, ql, logprior, ls), e[p + "ELBO"]) < TOL
        assert rel_err(O.q_probs(ql), e[p + "q"]) < TOL
        cf = O.elbo_grads_closed_form(act, y, True, N, O.q_probs(ql), log_std=ls, q_logits=ql, logprior=logprior,
                                      prior_loglikes=torch.zeros(D))
        assert rel_err(cf["dact"], e[p + "g_act"]) < 2e-5
        assert np.allclose(cf["dq_logits"].numpy(), e[p + "g_q_logits"], rtol=2e-4, atol=1e-4 * np.abs(e[p + "g_q_logits"]).max() + 1e-30)
        assert np.allclose(cf["dlog_std"].numpy(), e[p + "g_log_std"], rtol=2e-5, atol=1e-12)
        mu, std = O.marginalise_predict(act, O.q_probs(ql), get_std=True)
        assert rel_err(mu, e[p + "mu"]) < TOL
        assert np.allclose(std.numpy(), e[p + "std"], rtol=1e-4, atol=1e-6, equal_nan=True)
    assert int(e["o2_raises"]) == 1
    with pytest.raises(RuntimeError):
        O.joint_loglike(torch.zeros(3), torch.randn(3, 5, 2), torch.randn(5, 2), True, 10, torch.zeros(2))


@pytest.mark.parametrize("C", [10, 3, 100, 1000])
def test_edge_cases_classification(C):
    e = Case("edge")
    p = "cls_%d/" % C
    act, y, ql = e.t(p + "act"), e.t(p + "y").long(), e.t(p + "q_logits")
    N = int(e[p + "N"])
    D = act.shape[0]
    logprior = torch.full((D,), 1.0 / D).log()
    joint = O.joint_loglike(torch.zeros(D), act, y, False, N)
    assert rel_err(joint, e[p + "joint"]) < TOL
    assert rel_err(O.elbo(torch.zeros(D), act, y, False, N, ql, logprior), e[p + "ELBO"]) < TOL
    cf = O.elbo_grads_closed_form(act, y, False, N, O.q_probs(ql), q_logits=ql, logprior=logprior,
                                  prior_loglikes=torch.zeros(D))
    assert rel_err(cf["dact"], e[p + "g_act"]) < 2e-5
    assert rel_err(cf["dq_logits"], e[p + "g_q_logits"]) < 2e-4
    lm = O.marg_loglike(joint, logprior)
    assert rel_err(lm, e[p + "log_marg"]) < TOL
    post = O.depth_log_posterior(joint, logprior, lm).exp()
    cfm = O.elbo_grads_closed_form(act, y, False, N, post, vi=False)
    assert rel_err(cfm["dact"], e[p + "gm_act"]) < 2e-5
    assert rel_err(O.marginalise_predict(act, O.q_probs(ql), softmax=True), e[p + "probs"]) < TOL


@pytest.mark.parametrize("C", [10, 100, 3])
def test_label_semantics(C):
    """labels.npz: rows labelled -100 through the reference's CrossEntropyLoss / NLLLoss / pred.ne(y); labels outside
    [0, C) raise there and here."""
    e = Case("labels")
    p = "c%d/" % C
    act, y, ql, N = e.t(p + "act"), e.t(p + "y").long(), e.t(p + "q_logits"), int(e[p + "N"])
    D, B = act.shape[0], act.shape[1]
    assert int((y == -100).sum()) >= 2
    logprior = torch.full((D,), 1.0 / D).log()
    assert rel_err(O.loglike_per_act(act, y, False), e[p + "loglike"]) < TOL
    assert rel_err(O.joint_loglike(torch.zeros(D), act, y, False, N), e[p + "joint"]) < TOL
    assert rel_err(O.elbo(torch.zeros(D), act, y, False, N, ql, logprior), e[p + "ELBO"]) < TOL
    cf = O.elbo_grads_closed_form(act, y, False, N, O.q_probs(ql), q_logits=ql, logprior=logprior,
                                  prior_loglikes=torch.zeros(D))
    assert rel_err(cf["dact"], e[p + "g_act"]) < 2e-5
    assert rel_err(cf["dq_logits"], e[p + "g_q_logits"]) < 2e-4
    nll, err = O._train_metrics(act, y, O.q_probs(ql), False, None)
    assert abs(nll - float(e[p + "nll"])) < 2e-5 * abs(nll) and err == float(e[p + "err"])
    assert int(e[p + "raises_hi"]) == 1 and int(e[p + "raises_neg"]) == 1
    for bad in (C, -1):
        yb = y.clone()
        yb[0] = bad
        with pytest.raises(IndexError):
            O.loglike_per_act(act, yb, False)


def test_wiggle_generator_matches_reference():
    c = Case("c1_wiggle")
    X, Y = O.wiggle_data()
    assert np.array_equal(X, c["x"]) and np.array_equal(Y, c["y"])
