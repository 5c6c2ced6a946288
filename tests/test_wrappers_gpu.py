"""GPU parity of the drop-in API (models + probability module + DUN / DUN_VI wrappers) against the golden
vectors the real reference produced, and against an fp64 evaluation of the oracle where fp32 is noisy."""
import contextlib
import io

import numpy as np
import pytest
import torch

from oracle import dun_oracle as O
from tests.util import (Case, FC_CASES, PRIOR_CASES, model_prior_kwargs, oracle_grads_with_masks, rel_err,
                        stacked_grads_from_reference)

pytestmark = pytest.mark.gpu


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def build_net(c):
    """dun_amd wrapper in the exact state the golden case started from."""
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU")
    from dun_amd.probability import depth_categorical, depth_categorical_VI
    from dun_amd.stochastic_fc_models import arq_uncert_fc_MLP, arq_uncert_fc_resnet
    from dun_amd.training_wrappers import DUN, DUN_VI
    cfg = c.cfg
    ctor = arq_uncert_fc_MLP if cfg.get("mlp") else arq_uncert_fc_resnet
    model = ctor(cfg["din"], cfg["O"], cfg["W"], cfg["L"], **model_prior_kwargs(c))
    sd = {k: torch.from_numpy(np.array(v)) for k, v in c.sub("sd/").items()}
    model.load_state_dict(sd)
    if cfg["vi"]:
        pm = depth_categorical_VI(c.prior(), cuda=True)
        cls = DUN_VI
    else:
        pm = depth_categorical(c.prior(), cuda=True)
        cls = DUN
    net = quiet(cls, model, pm, cfg["N"], lr=cfg["lr"], momentum=cfg["momentum"], weight_decay=cfg["wd"],
                cuda=True, regression=cfg["regression"], pred_sig=cfg.get("pred_sig"))
    if cfg["vi"]:
        pm.q_logits.data.copy_(c.t("q_logits0"))
    if cfg["regression"]:
        net.f_neg_loglike.log_std.data.copy_(c.t("log_std0"))
    return net


def noise_bounded(got32, ref32, truth64, slack=4.0, floor=2e-5):
    return rel_err(got32, truth64) <= slack * rel_err(ref32, truth64) + floor


def oracle64(c):
    net = c.oracle_net(torch.float64)
    x = c.t("x").double()
    y = c.y() if not c.cfg["regression"] else c.y().double()
    return net, x, y


@pytest.mark.parametrize("name", FC_CASES + PRIOR_CASES)
def test_forward_and_objective(name):
    c = Case(name)
    cfg = c.cfg
    net = build_net(c)
    x, y = c.t("x").cuda(), c.y().cuda()
    net.model.train()
    act = net.model.forward(x)
    o64, x64, y64 = oracle64(c)
    p64 = O.clone_params(o64.p)
    act64 = O.fc_forward(p64, x64, train=True, residual=not cfg.get("mlp", False))
    assert noise_bounded(act.detach().cpu(), c["act_vec"], act64)
    # BatchNorm running statistics after one train-mode forward
    for i in range(cfg["L"]):
        assert rel_err(net.model.running_mean[i].cpu(), p64["running_mean"][i]) < 1e-5
        assert rel_err(net.model.running_var[i].cpu(), p64["running_var"][i]) < 1e-5
    assert int(net.model.num_batches_tracked.sum()) == cfg["L"]
    prior = net.model.get_w_prior_loglike(k=None)
    if cfg["vi"]:
        out, ll = net.prob_model.fused_objective(prior, act, y, net.f_neg_loglike, cfg["N"], want_ll=True)
        assert rel_err(out.objective.cpu(), c["ELBO"]) < 1e-5
        assert rel_err(out.weights.cpu(), c["q"]) < 1e-5
        assert abs(float(out.KL) - float(c["KL"])) < 1e-5
        assert rel_err(out.E_loglike.cpu(), c["E_loglike"]) < 1e-5
        # the reference-named differentiable entry gives the same number
        elbo = net.prob_model.estimate_ELBO(prior, act.detach(), y, net.f_neg_loglike, cfg["N"], Beta=1)
        assert rel_err(elbo.detach().cpu(), c["ELBO"]) < 1e-5
    else:
        out, ll = net.prob_model.fused_objective(prior, act, y, net.f_neg_loglike, cfg["N"], want_ll=True)
        assert rel_err(out.objective.cpu(), c["log_marg"]) < 1e-5
    # north-star parity bar: per-depth log-likelihoods within 1e-5 relative of the reference
    assert rel_err(out.joint.cpu(), c["joint"]) < 1e-5
    assert rel_err(ll.cpu().sum(dim=1), c["loglike"].sum(axis=1)) < 1e-5
    ll64 = O.loglike_per_act(act64, y64, cfg["regression"], o64.log_std)
    assert noise_bounded(ll.cpu(), c["loglike"], ll64)
    assert rel_err(out.log_marg.cpu(), c["log_marg"]) < 1e-5
    assert np.allclose(out.posterior.cpu().numpy(), np.exp(c["log_post"].astype(np.float64)), rtol=5e-3, atol=1e-6)
    jl = net.prob_model.get_w_joint_loglike(prior, act.detach(), y, net.f_neg_loglike, cfg["N"])
    assert rel_err(jl.detach().cpu(), c["joint"]) < 1e-5


@pytest.mark.parametrize("name", FC_CASES + PRIOR_CASES)
def test_gradients(name):
    c = Case(name)
    cfg = c.cfg
    net = build_net(c)
    x, y = c.t("x").cuda(), c.y().cuda()
    net.model.train()
    act = net.model.forward(x)
    prior = net.model.get_w_prior_loglike(k=None)
    out = net.prob_model.fused_objective(prior, act, y, net.f_neg_loglike, cfg["N"])
    o64, x64, y64 = oracle64(c)
    o64b = c.oracle_net(torch.float64)
    _, g64, act64, joint64 = o64.fit(x64, y64, return_grads=True)
    w64 = O.q_probs(o64b.q_logits) if cfg["vi"] else O.depth_log_posterior(joint64, o64b.logprior).exp()
    dact64 = O.elbo_grads_closed_form(act64, y64, cfg["regression"], cfg["N"], w64, log_std=o64b.log_std,
                                      vi=False)["dact"]
    assert noise_bounded(out.dact.cpu().view(c["g/act_vec"].shape), c["g/act_vec"], dact64)
    net._backward(act, prior, out)
    ref = stacked_grads_from_reference(c)
    # truth for OUR gradients: fp64 with our ReLU activation pattern (see oracle_grads_with_masks); the error
    # budget stays the reference's own fp32 error against the unconstrained fp64 evaluation
    buf = list(net.model._fc_cache.values())[0]
    masks = (buf.Z > 0).cpu()
    gm64, _ = oracle_grads_with_masks(c, masks)

    def bounded(got, ref32, key, slack=6.0, floor=2e-5):
        return rel_err(got, gm64[key]) <= slack * rel_err(ref32, g64[key]) + floor

    for k, v in ref.items():
        assert bounded(net.model.stacked_grad[k].cpu(), v, k), k
    if cfg["vi"]:
        assert bounded(net.prob_model.q_logits.grad.cpu(), c["g/q_logits"], "q_logits")
    if cfg["regression"] and cfg.get("pred_sig") is None:
        assert bounded(net.f_neg_loglike.log_std.grad.cpu(), c["g/log_std"], "log_std")
    # the autograd route (estimate_ELBO / get_marg_loglike + loss.backward()) lands on the same gradients
    net2 = build_net(c)
    net2.model.train()
    act2 = net2.model.forward(x)
    prior2 = net2.model.get_w_prior_loglike(k=None)
    if cfg["vi"]:
        obj = net2.prob_model.estimate_ELBO(prior2, act2, y, net2.f_neg_loglike, cfg["N"], Beta=1)
    else:
        joint = net2.prob_model.get_w_joint_loglike(prior2, act2, y, net2.f_neg_loglike, cfg["N"])
        obj = net2.prob_model.get_marg_loglike(joint)
    (-obj / cfg["N"]).backward()
    for k in ref:
        a, b = net2.model.stacked_grad[k].cpu(), net.model.stacked_grad[k].cpu()
        assert rel_err(a, b) < 1e-4, k  # same kernels and inputs; depth weights come from torch's fp32 softmax here
    if cfg["vi"]:
        assert rel_err(net2.prob_model.q_logits.grad.cpu(), net.prob_model.q_logits.grad.cpu()) < 1e-4
    if cfg["regression"] and cfg.get("pred_sig") is None:
        assert rel_err(net2.f_neg_loglike.log_std.grad.cpu(), net.f_neg_loglike.log_std.grad.cpu()) < 1e-4


@pytest.mark.parametrize("name", FC_CASES + PRIOR_CASES)
def test_fit_eval_predict(name):
    c = Case(name)
    cfg = c.cfg
    net = build_net(c)
    x, y = c.t("x"), c.y()
    for step in range(3):
        ret = net.fit(x, y)
        # step 0 is a pure function of the inputs; later steps carry fp32 noise through the update (including
        # O(1/B) gradient changes from ReLU sign flips at pre-activations within rounding noise of zero)
        rt = 2e-5 if step == 0 else 1e-2
        want = c["fit/ret"][step]
        keep = [0, 2] if (np.isnan(ret[1]) and not cfg["vi"]) else [0, 1, 2]
        # (exact-posterior runs: a one-hot posterior puts the predictive variance within rounding of 0, where
        #  sqrt(negative) = NaN is the reference's own semantics and which side of 0 it lands on is a coin toss)
        assert np.allclose(np.array(ret)[keep], want[keep], rtol=rt, atol=1e-6), (step, ret, want)
        assert np.allclose(net.prob_model.current_posterior.cpu().numpy(), c["fit/posterior"][step], rtol=1e-2,
                           atol=1e-6)
    sd = c.sub("fit/sd/")
    got = net.model.state_dict()
    for k in ("input_layer.weight", "output_layer.weight", "output_layer.bias"):
        assert rel_err(got[k].cpu(), sd[k]) < 1e-2, k
    for i in range(cfg["L"]):
        for suffix in ("block.0.weight", "block.2.weight", "block.2.running_mean", "block.2.running_var"):
            k = "stochstic_layers.%d.%s" % (i, suffix)
            assert rel_err(got[k].cpu(), sd[k]) < 1e-2, k
        assert int(got["stochstic_layers.%d.block.2.num_batches_tracked" % i]) == 3
    if cfg["vi"]:
        assert rel_err(net.prob_model.q_logits.data.cpu(), c["fit/q_logits"]) < 1e-2
    if cfg["regression"]:
        assert rel_err(net.f_neg_loglike.log_std.data.cpu(), c["fit/log_std"]) < 1e-2
    ev, want_ev = np.array(net.eval(x, y)), c["eval/ret"]
    keep = [1] if (np.isnan(ev[0]) and not cfg["vi"]) else [0, 1]
    assert np.allclose(ev[keep], want_ev[keep], rtol=1e-2)
    xt = c.t("x_test")
    assert rel_err(net.layer_predict(xt).cpu(), c["layer_predict"]) < 1e-2
    if cfg["regression"]:
        mu, std = net.predict(xt, get_std=True)
        assert rel_err(mu.cpu(), c["pred/mu"]) < 1e-2
        both = ~(np.isnan(std.cpu().numpy()) | np.isnan(c["pred/std"]))
        assert both.mean() > 0.9 and rel_err(std.cpu().numpy()[both], c["pred/std"][both]) < 1e-2
        _, mstd = net.predict(xt, get_std=True, return_model_std=True)
        both = ~(np.isnan(mstd.cpu().numpy()) | np.isnan(c["pred/model_std"]))
        assert both.mean() > 0.9
        # (a std near 0 is the square root of rounding noise of size eps*mu^2: only its scale is comparable)
        if cfg["vi"]:
            assert np.allclose(mstd.cpu().numpy()[both], c["pred/model_std"][both], rtol=2e-2,
                               atol=2e-3 * float(np.abs(c["pred/mu"]).max()))
        else:
            # exact posterior ~ one-hot: the model std is sqrt(post_2) * |a_2 - a_1| with post_2 ~ 1e-5 depending
            # exponentially on the per-depth joint; after three noisy steps only its order of magnitude is pinned
            assert float(np.nanmax(mstd.cpu().numpy())) < 10 * float(np.nanmax(c["pred/model_std"])) + 1e-6
        assert rel_err(net.predict(xt).cpu(), c["pred/plain"]) < 1e-2
        if c.has("pred/depth_k"):
            k = int(c["pred/depth_k"])
            net.prob_model.current_posterior = c.t("pred/depth_k_q").cuda()
            mu, _ = net.predict(xt, depth=k, get_std=True, return_model_std=True)
            assert rel_err(mu.cpu(), c["pred/depth_k_mu"]) < 1e-2
    else:
        assert rel_err(net.predict(xt).cpu(), c["pred/plain"]) < 1e-2
    if cfg.get("sweep"):
        bs, N = int(c["sweep/batch"]), cfg["N"]
        ds = torch.utils.data.TensorDataset(x[:N], y[:N])
        loader = torch.utils.data.DataLoader(ds, batch_size=bs, shuffle=False)
        post, lm = net.get_exact_d_posterior(loader, train_bn=False, logposterior=False)
        assert np.allclose(post.cpu().numpy(), c["sweep/posterior"], rtol=2e-3, atol=1e-6)
        assert abs(lm - float(c["sweep/log_marg"])) < 1e-2 * abs(lm)
        if cfg["vi"]:
            assert abs(net.get_exact_ELBO(loader) - float(c["sweep/ELBO"])) < 1e-2 * abs(float(c["sweep/ELBO"]))


def test_seed_matched_construction_and_checkpoint(tmp_path):
    """Same torch seed + same construction order as the reference => same initial weights (c1 fixture holds the
    reference's state_dict for seed 0, with BN affine/running stats overwritten afterwards)."""
    from dun_amd.stochastic_fc_models import arq_uncert_fc_resnet
    c = Case("c1_wiggle")
    torch.manual_seed(0)
    m = arq_uncert_fc_resnet(1, 1, 100, 10)
    sd = c.sub("sd/")
    got = m.state_dict()
    for k in ("input_layer.weight", "input_layer.bias", "output_layer.weight", "stochstic_layers.9.block.0.weight"):
        assert np.array_equal(got[k].numpy(), sd[k]), k
    assert set(got.keys()) == set(sd.keys())
    net = build_net(c)
    net.fit(c.t("x"), c.y())
    f = str(tmp_path / "theta.dat")
    quiet(net.save, f)
    net2 = build_net(c)
    quiet(net2.load, f)
    a, b = net.predict(c.t("x")), net2.predict(c.t("x"))
    assert torch.equal(a, b)
    r1, r2 = net.fit(c.t("x"), c.y()), net2.fit(c.t("x"), c.y())
    assert np.allclose(r1, r2, rtol=1e-6)


def test_reference_state_checkpoint_on_the_device(tmp_path):
    """tests/test_checkpoint_cpu.py on the real kernels: the state written from the reference's objects is ingested into a
    device wrapper (per-parameter momentum scattered into the flat buffer of the fused SGD kernel, schedule restored)
    and the continued training -- graphed steps included: the net is small -- equals the reference's; a round trip
    through save_state / load_state on the device continues bit-identically."""
    import os
    from tests.test_checkpoint_cpu import build, continue_like_reference
    from tests.util import GOLDEN
    c = Case("ckpt_ref_vi")
    net = build(c.cfg, seed=99, cuda=True)
    x, y = c.t("x").cuda(), c.t("y").cuda()
    for _ in range(3):
        net.fit(x, y)  # eager, capture, replay: the captured step must not survive the load
    assert quiet(net.load_state, os.path.join(GOLDEN, "ckpt_ref_vi.pt")) == (1, 0.25)
    continue_like_reference(net, c, dev="cuda")
    path = str(tmp_path / "state.pt")
    quiet(net.save_state, path)
    other = build(c.cfg, seed=5, cuda=True)
    assert quiet(other.load_state, path) == net.epoch
    for _ in range(3):
        assert net.fit(x, y) == other.fit(x, y)
    assert torch.equal(net.model.flat_params, other.model.flat_params)


def test_one_row_batches():
    """A training batch of one row raises BatchNorm's ValueError like the reference; one-row evaluation runs the
    kernels at B = 1 and equals the same row inside a larger batch."""
    for name in ("c2_boston", "c3_reduced", "cls_fc"):
        c = Case(name)
        net = build_net(c)
        x, y = c.t("x").cuda(), c.y().cuda()
        net.fit(x, y)
        with pytest.raises(ValueError, match="more than 1 value per channel"):
            net.fit(x[:1], y[:1])
        full = net.predict(x[:5])
        one = net.predict(x[3:4])
        assert one.shape == full[3:4].shape
        assert torch.allclose(one, full[3:4], rtol=2e-5, atol=1e-6), name
        ev = net.eval(x[:1], y[:1])
        assert all(np.isfinite(v) for v in ev), (name, ev)
        none = net.predict(x[:0])  # an empty inference batch gives empty outputs, like the reference's torch ops
        assert none.shape == (0,) + tuple(full.shape[1:]) and none.is_cuda
        assert np.allclose(net.fit(x, y), c["fit/ret"][1], rtol=2e-3, atol=1e-6)  # the refused step left no trace


def test_label_semantics_through_the_wrapper():
    from tests.test_host_cpu import check_label_semantics_through_the_wrapper
    check_label_semantics_through_the_wrapper(build_net)


def test_predict_with_truncated_depth_needs_a_matching_posterior():
    """predict(x, depth=k) with the full-length posterior: the reference's (q * preds) broadcast raises; so do we
    (the kernel would otherwise use an unnormalised prefix of q)."""
    c = Case("c2_boston")
    net = build_net(c)
    net.fit(c.t("x"), c.y())
    with pytest.raises(RuntimeError):
        net.predict(c.t("x"), depth=3)


@pytest.mark.parametrize("vi", [True, False])
@pytest.mark.parametrize("din,W,L,B", [(64, 256, 3, 1024), (16, 2048, 2, 640), (7, 768, 2, 333), (13, 100, 10, 512),
                                       (5, 257, 2, 129), (3, 512, 0, 64), (8, 1024, 1, 40000)])
def test_likelihood_pass_inside_the_head_matches_the_separate_pass(din, W, L, B, vi):
    """forward(x, objective=...) -- BatchNorm apply + residual + head + Gaussian likelihood terms in one row-owned pass
    (widths that are multiples of 256), or the head kernel with the likelihood epilogue (any other width, small nets)
    -- against forward(x) followed by the streaming pass over act_vec: same act_vec, dact, sums, objective; train and
    eval mode; and a whole fit step either way lands on the same parameters."""
    from dun_amd.probability import depth_categorical, depth_categorical_VI
    from dun_amd.stochastic_fc_models import arq_uncert_fc_resnet
    from dun_amd.training_wrappers import DUN, DUN_VI
    D, N = L + 1, 5000

    def make():
        torch.manual_seed(W + L)
        model = arq_uncert_fc_resnet(din, 1, W, L)
        pm = (depth_categorical_VI if vi else depth_categorical)([1.0 / D] * D, cuda=True)
        net = quiet(DUN_VI if vi else DUN, model, pm, N, lr=1e-3, momentum=0.5, cuda=True, regression=True)
        g = torch.Generator().manual_seed(B)
        if vi:
            pm.q_logits.data.copy_(0.3 * torch.randn(D, generator=g))
        net.f_neg_loglike.log_std.data.fill_(0.2)
        return net, torch.randn(B, din, generator=g).cuda(), torch.randn(B, 1, generator=g).cuda()

    net, x, y = make()
    for train in (True, False):
        net.model.train(train)
        rm = net.model.running_mean.clone()
        with torch.no_grad():
            act_plain = net.model.forward(x).clone()
            net.model.running_mean.copy_(rm)  # (a train-mode forward moves the running statistics)
            plain = net.prob_model.fused_objective(None, act_plain, y, net.f_neg_loglike, N)
            net.model.running_mean.copy_(rm)
            spec = net.prob_model.head_objective(y, net.f_neg_loglike, B)
            assert spec is not None
            act_fused = net.model.forward(x, objective=spec)
            fused = net.prob_model.fused_objective(None, act_fused, y, net.f_neg_loglike, N, head_pass=spec)
        assert spec.partials is not None and (spec.dact is not None) == vi
        assert rel_err(act_fused.cpu(), act_plain.cpu()) < 2e-6
        assert rel_err(fused.sums.cpu(), plain.sums.cpu()) < 1e-6
        assert rel_err(fused.joint.cpu(), plain.joint.cpu()) < 1e-6
        assert rel_err(fused.objective.cpu(), plain.objective.cpu()) < 1e-6
        assert rel_err(fused.dact.cpu().view(-1), plain.dact.cpu().view(-1)) < 2e-5
    # whole steps: the wrapper takes the fused route by itself; switch it off for the other net
    a, xa, ya = make()
    b, _, _ = make()
    b.model.supports_head_objective = False
    for _ in range(2):
        ra, rb = a.fit(xa, ya), b.fit(xa, ya)
        assert np.allclose(ra, rb, rtol=2e-5, equal_nan=True), (ra, rb)
        # the head's weight / bias gradient: accumulated by the forward's head kernels (a, VI on fused widths) against
        # dun_head_backward's pass over the hidden states (b)
        for k in ("w_out", "b_out", "w_in"):
            assert rel_err(a.model.stacked_grad[k].cpu(), b.model.stacked_grad[k].cpu()) < 2e-5, k
    if vi and L > 0 and W % 256 == 0:
        buf = [v for key, v in a.model._fc_cache.items() if key[2]][0]
        assert buf.c.head_grads_from_forward == 1 and buf.c.obj_dwpart is not None
    assert rel_err(a.model.flat_params.cpu(), b.model.flat_params.cpu()) < 1e-5
    assert np.allclose(a.eval(xa, ya), b.eval(xa, ya), rtol=1e-5, equal_nan=True)


def test_o2_regression_raises_like_the_reference():
    from dun_amd.probability import depth_categorical, homo_Gauss_mloglike
    with pytest.raises(RuntimeError):
        depth_categorical.get_w_joint_loglike(torch.zeros(3).cuda(), torch.randn(3, 5, 2).cuda(),
                                              torch.randn(5, 2).cuda(), homo_Gauss_mloglike(2).cuda(), 10)


def test_sharded_path_world1_matches_unsharded():
    """The staged C entries + torch.distributed(nccl = RCCL) plumbing at world size 1 must give exactly the numbers of
    the fused single-call path (same kernels, same order)."""
    import socket
    import torch.distributed as dist
    from dun_amd.distributed import shard
    c = Case("c3_reduced")
    x, y = c.t("x"), c.y()
    ref = build_net(c)
    want = [ref.fit(x, y) for _ in range(2)]
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        net = build_net(c)
        quiet(shard, net, sync_bn=True)
        got = [net.fit(x, y) for _ in range(2)]
        assert np.allclose(got, want, rtol=1e-6)
        # the same two steps with every collective of the N-GPU step ISSUED on the one-rank communicator: asynchronous
        # BatchNorm exchange with the deferred dW GEMM underneath, per-block gradient messages, likelihood sums
        net2 = build_net(c)
        quiet(shard, net2, sync_bn=True).force_collectives = True
        got2 = [net2.fit(x, y) for _ in range(2)]
        assert np.allclose(got2, want, rtol=1e-6)
        assert torch.equal(net2.model.flat_params, net.model.flat_params)
        net3 = build_net(c)  # forward exchanges asynchronous as well (off by default)
        spec3 = quiet(shard, net3, sync_bn=True)
        spec3.force_collectives = spec3.async_forward_bn = True
        assert np.allclose([net3.fit(x, y) for _ in range(2)], want, rtol=1e-6)
        assert torch.equal(net3.model.flat_params, net.model.flat_params)
        # the collectives the N-GPU path issues, through RCCL itself (Shard skips them at world size 1): fp64 SUM,
        # asynchronous SUM on a slice of the flat gradient buffer, MAX, broadcast, barrier
        t = torch.arange(128, dtype=torch.float64, device="cuda").view(2, 64)
        dist.all_reduce(t)
        piece = net.model.flat_grads[8:1000]
        before = piece.clone()
        h = dist.all_reduce(piece, op=dist.ReduceOp.SUM, async_op=True)
        h.wait()
        m = torch.tensor([1.5], dtype=torch.float64, device="cuda")
        dist.all_reduce(m, op=dist.ReduceOp.MAX)
        dist.broadcast(net.model.flat_params, src=0)
        dist.barrier()
        torch.cuda.synchronize()
        assert torch.equal(t.view(-1), torch.arange(128, dtype=torch.float64, device="cuda"))
        assert torch.equal(piece, before) and float(m) == 1.5
        assert torch.equal(net.model.flat_params, ref.model.flat_params)
        assert torch.equal(net.model.running_var, ref.model.running_var)
        assert np.allclose(net.eval(x, y), ref.eval(x, y), rtol=1e-6)
        # image DUN: every BatchNorm2d call through dun_amd/sync_bn.py with its fp64 [2,C] exchange issued on RCCL
        # (one rank: the sums come back unchanged) against MIOpen's own BatchNorm in the unsharded step
        from tests.test_image_dun import build as build_img
        ci = Case("img_r18")
        xi, yi = ci.t("x"), ci.t("y").long()
        ref_i, _ = build_img(ci, cuda=True)
        want_i = [ref_i.fit(xi, yi) for _ in range(2)]
        net_i, _ = build_img(ci, cuda=True)
        spec_i = quiet(shard, net_i, sync_bn=True)
        spec_i.force_collectives = True
        assert spec_i.sync_bn2d_modules > 0
        got_i = [net_i.fit(xi, yi) for _ in range(2)]
        assert np.allclose(got_i, want_i, rtol=2e-5, atol=1e-7), (got_i, want_i)
        for (k, a), b in zip(net_i.model.state_dict().items(), ref_i.model.state_dict().values()):
            scale = max(float(b.abs().max()), 1e-30) if b.is_floating_point() else 1.0
            assert float((a.double() - b.double()).abs().max()) <= 2e-5 * scale + 1e-7, k
    finally:
        dist.destroy_process_group()


def test_discarded_net_releases_its_buffers(monkeypatch):
    """model -> buffers -> act_vec -> grad_fn -> parameters -> model would be a cycle through the C++ autograd graph
    (invisible to Python's collector); the buffers do not keep the autograd output, so a trained-and-dropped net gives
    its activation memory back."""
    import gc
    monkeypatch.setenv("DUN_NO_GRAPH", "1")
    c = Case("c3_reduced")
    x, y = c.t("x").cuda(), c.y().cuda()
    gc.collect()
    torch.cuda.synchronize()
    before = torch.cuda.memory_allocated()
    net = build_net(c)
    net.fit(x, y)
    net.fit(x, y)
    held = torch.cuda.memory_allocated() - before
    del net
    gc.collect()
    torch.cuda.synchronize()
    left = torch.cuda.memory_allocated() - before
    assert held > 0 and left < 0.05 * held, (held, left)


@pytest.mark.parametrize("name", ["c2_boston", "cls_fc", "mll_small"])
def test_graphed_step_is_bit_identical_to_eager(name, monkeypatch):
    """The whole-step hipGraph (captured on the 2nd call) must reproduce the eager launch sequence exactly, also
    across an lr change (re-capture) and with eval/predict calls in between."""
    c = Case(name)
    x, y = c.t("x").cuda(), c.y().cuda()
    monkeypatch.setenv("DUN_NO_GRAPH", "1")
    eager = build_net(c)
    want = [eager.fit(x, y) for _ in range(4)]
    eager.optimizer.param_groups[0]["lr"] *= 0.1
    want += [eager.fit(x, y) for _ in range(3)]
    want_eval = eager.eval(x, y)
    monkeypatch.delenv("DUN_NO_GRAPH")
    net = build_net(c)
    got = []
    for i in range(4):
        got.append(net.fit(x + 0 if i % 2 else x, y))      # fresh input tensors: the graph owns static copies
        if i == 2:
            net.predict(c.t("x_test"))                     # eager inference between replays
    assert net._graph_state["graph"] is not None
    net.optimizer.param_groups[0]["lr"] *= 0.1
    got += [net.fit(x, y) for _ in range(3)]
    assert np.array_equal(np.array(got), np.array(want), equal_nan=True), (got, want)
    assert torch.equal(net.model.flat_params, eager.model.flat_params)
    assert torch.equal(net.model.running_var, eager.model.running_var)
    assert torch.equal(net.model.num_batches_tracked, eager.model.num_batches_tracked)
    assert torch.equal(net.prob_model.current_posterior, eager.prob_model.current_posterior)
    assert np.array_equal(np.array(net.eval(x, y)), np.array(want_eval), equal_nan=True)


def test_graphed_steps_with_a_ragged_last_batch(monkeypatch):
    """A loader whose last batch is shorter alternates between two step configurations: each gets its own captured
    graph (kept side by side, replayed from the third epoch on) and the trajectory equals the eager one bit for bit."""
    c = Case("c2_boston")
    x, y = c.t("x").cuda(), c.y().cuda()
    batches = [(x[:200], y[:200]), (x[200:400], y[200:400]), (x[400:], y[400:])]   # 200, 200, 112 rows
    monkeypatch.setenv("DUN_NO_GRAPH", "1")
    eager = build_net(c)
    want = [eager.fit(xb, yb) for _ in range(4) for xb, yb in batches]
    monkeypatch.delenv("DUN_NO_GRAPH")
    net = build_net(c)
    got = [net.fit(xb, yb) for _ in range(4) for xb, yb in batches]
    assert np.array_equal(np.array(got), np.array(want), equal_nan=True)
    assert len(net._graph_states) == 2 and all(st["graph"] is not None for st in net._graph_states.values())
    assert torch.equal(net.model.flat_params, eager.model.flat_params)
    assert torch.equal(net.prob_model.current_posterior, eager.prob_model.current_posterior)


def test_captured_steps_outlive_the_models_buffer_cache(monkeypatch):
    """More distinct batch sizes than either cache holds (four captured steps per wrapper, four buffer sets per model,
    evicted on different schedules): a captured step keeps the activation buffers it was captured on alive, so
    returning to the first batch size replays onto valid memory -- the trajectory equals the eager one bit for bit."""
    c = Case("c2_boston")
    x, y = c.t("x").cuda(), c.y().cuda()
    sizes = [64, 80, 96, 112, 128, 144]
    order = [s for s in sizes for _ in range(2)] + [64, 64, 80, 144, 64]
    monkeypatch.setenv("DUN_NO_GRAPH", "1")
    eager = build_net(c)
    want = [eager.fit(x[:n], y[:n]) for n in order]
    monkeypatch.delenv("DUN_NO_GRAPH")
    net = build_net(c)
    got = []
    for n in order:
        got.append(net.fit(x[:n], y[:n]))
        junk = torch.randn(1 << 16, device="cuda")  # whatever the allocator hands out next is overwritten
        del junk
    assert np.array_equal(np.array(got), np.array(want), equal_nan=True), (got, want)
    assert torch.equal(net.model.flat_params, eager.model.flat_params)
    for st in net._graph_states.values():
        if st["graph"] is not None:
            assert len(st["keep"]) > 0


def test_train_fc_epoch_loop_on_device_loader(tmp_path):
    """Config-1 end to end on the GPU: the reference-shaped epoch loop over a device-resident loader."""
    from dun_amd.datasets import DeviceLoader, load_wiggle
    from dun_amd.probability import depth_categorical_VI
    from dun_amd.stochastic_fc_models import arq_uncert_fc_resnet
    from dun_amd.train_fc import train_fc_DUN
    from dun_amd.training_wrappers import DUN_VI
    X, Y = load_wiggle()
    train = DeviceLoader(X, Y, batch_size=128, shuffle=True, device="cuda", seed=0)
    val = DeviceLoader(X, Y, batch_size=128, device="cuda")
    torch.manual_seed(0)
    model = arq_uncert_fc_resnet(1, 1, 100, 10)
    net = quiet(DUN_VI, model, depth_categorical_VI([1.0 / 11] * 11, cuda=True), 300, lr=1e-3, momentum=0.9,
                weight_decay=1e-4, cuda=True, regression=True)
    out = quiet(train_fc_DUN, net, "t", str(tmp_path), 128, 12, train, val, True, seed=0, nb_its_dev=3,
                track_posterior=True, track_exact_ELBO=True, save_freq=6, q_nograd_its=2)
    elbo, nll_tr, nll_dev, err_tr, err_dev, q_app, q_true, lik, exact, basedir = out
    assert np.all(np.isfinite(elbo)) and elbo[-1] > elbo[0]
    assert np.all(np.isfinite(exact)) and exact[-1] > exact[0]
    assert err_tr[-1] < err_tr[0]
    assert q_app.shape == (12, 11) and abs(q_app[-1].sum() - 1) < 1e-5 and abs(q_true[-1].sum() - 1) < 1e-4


@pytest.mark.parametrize("din,W,L,B,C", [(3, 37, 2, 77, 0), (5, 50, 3, 129, 7), (1, 130, 1, 2, 0), (9, 257, 2, 513, 3),
                                          (4, 64, 0, 33, 0)])
def test_odd_shapes_against_fp64_oracle_on_device(din, W, L, B, C):
    """Widths / batches that are not multiples of anything (scalar-load kernels, ragged tiles, L = 0), regression
    (C = 0) and classification: forward, objective and every gradient against the oracle in fp64 on the device."""
    from dun_amd.probability import depth_categorical_VI
    from dun_amd.stochastic_fc_models import arq_uncert_fc_resnet
    from dun_amd.training_wrappers import DUN_VI
    regression = C == 0
    O_dim = 1 if regression else C
    D, N = L + 1, 1000
    torch.manual_seed(din * 1000 + W)
    model = arq_uncert_fc_resnet(din, O_dim, W, L)
    net = quiet(DUN_VI, model, depth_categorical_VI([1.0 / D] * D, cuda=True), N, cuda=True, regression=regression)
    g = torch.Generator().manual_seed(B)
    net.prob_model.q_logits.data.copy_(0.3 * torch.randn(D, generator=g))
    x = torch.randn(B, din, generator=g).cuda()
    y = (torch.randn(B, 1, generator=g) if regression else torch.randint(0, C, (B,), generator=g)).cuda()
    net.model.train()
    act = net.model.forward(x)
    prior = net.model.get_w_prior_loglike(k=None)
    out = net.prob_model.fused_objective(prior, act, y, net.f_neg_loglike, N)
    net._backward(act, prior, out)
    dt = torch.float64
    p = {k: net.model.stacked[k].detach().to(dt).clone().requires_grad_(True) for k in O.PARAM_KEYS}
    p["running_mean"] = torch.zeros(L, W, device="cuda", dtype=dt)
    p["running_var"] = torch.ones(L, W, device="cuda", dtype=dt)
    p["num_batches_tracked"] = torch.zeros(L, dtype=torch.int64, device="cuda")
    ql = net.prob_model.q_logits.detach().to(dt).clone().requires_grad_(True)
    ls = net.f_neg_loglike.log_std.detach().to(dt).clone().requires_grad_(True) if regression else None
    logprior = torch.full((D,), 1.0 / D, device="cuda", dtype=dt).log()
    a64 = O.fc_forward(p, x.to(dt), train=True)
    joint = O.joint_loglike(torch.zeros(D, device="cuda", dtype=dt), a64, y.to(dt) if regression else y, regression, N, ls)
    elbo = (O.q_probs(ql) * joint).sum() - O.kl_q_prior(ql, logprior)
    leaves = [p[k] for k in O.PARAM_KEYS if p[k].numel()] + [ql] + ([ls] if regression else [])
    names = [k for k in O.PARAM_KEYS if p[k].numel()] + ["q_logits"] + (["log_std"] if regression else [])
    g64 = dict(zip(names, torch.autograd.grad(-elbo / N, leaves)))
    assert rel_err(act.detach().cpu(), a64.detach().cpu()) < 2e-5
    assert rel_err(out.joint.cpu(), joint.detach().cpu()) < 1e-5
    assert rel_err(out.objective.cpu(), elbo.detach().cpu()) < 1e-5
    for k in names:
        mine = (net.prob_model.q_logits.grad if k == "q_logits" else
                net.f_neg_loglike.log_std.grad if k == "log_std" else net.model.stacked_grad[k])
        ref = g64[k]
        err = float((mine.double() - ref).norm() / ref.norm().clamp_min(1e-30))
        assert err < 2e-3, (k, err)   # l2 over the tensor: a ReLU sign flip near 0 moves single entries


def test_vec_forward_and_vec_predict_cold_path():
    """vec_forward / vec_predict (plotting-only in the reference: torch ops over the same parameters): an all-ones mask
    is the deepest exit, a prefix mask is that depth's exit, both equal to the kernel path's eval forward and to the
    oracle's fc_vec_forward."""
    c = Case("cls_fc")
    net = build_net(c)
    x = c.t("x").cuda()
    net.model.eval()
    L = c.cfg["L"]
    with torch.no_grad():
        act = net.model.forward(x)                      # [D, B, C] through the HIP kernels
        full = net.model.vec_forward(x, torch.ones(L, device="cuda"))
        assert rel_err(full.cpu(), act[-1].cpu()) < 1e-5
        for k in range(L + 1):
            mask = torch.cat([torch.ones(k), torch.zeros(L - k)]).cuda()
            assert rel_err(net.model.vec_forward(x, mask).cpu(), act[k].cpu()) < 1e-5
        p = O.params_from_reference_state_dict({k: v.cpu() for k, v in net.model.state_dict().items()}, L)
        mask = torch.tensor([1.0, 0.0, 1.0, 0.0][:L])
        ref = O.fc_vec_forward(p, c.t("x"), mask)
        assert rel_err(net.model.vec_forward(x, mask.cuda()).cpu(), ref) < 1e-5
    bin_mat = torch.tensor([[1.0] * L, [1.0] + [0.0] * (L - 1)], device="cuda")
    probs = net.vec_predict(x, bin_mat)
    assert probs.shape == (2, x.shape[0], c.cfg["O"])
    assert rel_err(probs[0].cpu(), torch.softmax(act[-1], dim=1).cpu()) < 1e-5
    assert rel_err(probs[1].cpu(), torch.softmax(act[1], dim=1).cpu()) < 1e-5
