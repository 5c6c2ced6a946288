#!/usr/bin/env python
"""Generate golden vectors by running the REAL reference (read-only at /root/reference) on CPU.

Run in the build container only:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
Outputs tests/golden/*.npz (data only: seeded inputs, weights, and what the reference computed).
The reference never travels to the GPU box; tests read only these files.
"""
import contextlib
import copy
import io
import json
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
sys.path.insert(0, REF)
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))

with contextlib.redirect_stdout(io.StringIO()):
    from src.probability import depth_categorical, depth_categorical_VI, homo_Gauss_mloglike  # noqa: E402
    from src.probability import diag_w_Gauss_loglike  # noqa: E402
    from src.DUN.training_wrappers import DUN, DUN_VI  # noqa: E402
    from src.DUN.stochastic_fc_models import arq_uncert_fc_resnet, arq_uncert_fc_MLP  # noqa: E402
    from src.datasets.additional_gap_loader import load_wiggle  # noqa: E402

torch.set_num_threads(4)


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def npy(t):
    return t.detach().cpu().numpy().copy()


def put_sd(out, prefix, sd):
    for k, v in sd.items():
        out[prefix + k] = npy(v)


def synthetic_regression(B, din, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, din, generator=g)
    w = torch.randn(din, 1, generator=g) / np.sqrt(din)
    y = torch.tanh(x @ w) + 0.1 * torch.randn(B, 1, generator=g)
    return x, y


def build(cfg):
    torch.manual_seed(cfg["seed"])
    ctor = arq_uncert_fc_MLP if cfg.get("mlp") else arq_uncert_fc_resnet
    w_prior = None
    if cfg.get("w_prior"):  # weight prior of the reference's regression experiments (src/probability.py:30-40)
        wp = cfg["w_prior"]
        w_prior = diag_w_Gauss_loglike(wp["mu"], wp["sig2"], wp["mult"])
    model = ctor(cfg["din"], cfg["O"], cfg["W"], cfg["L"], w_prior=w_prior, BMA_prior=bool(cfg.get("bma")))
    D = cfg["L"] + 1
    prior = cfg.get("prior") or [1.0 / D] * D
    if cfg["vi"]:
        pm = quiet(depth_categorical_VI, prior, cuda=False)
        cls = DUN_VI
    else:
        pm = quiet(depth_categorical, prior, cuda=False)
        cls = DUN
    net = quiet(cls, model, pm, cfg["N"], lr=cfg["lr"], momentum=cfg["momentum"], weight_decay=cfg["wd"],
                cuda=False, regression=cfg["regression"], pred_sig=cfg.get("pred_sig"))
    g = torch.Generator().manual_seed(cfg["seed"] + 1000)
    if cfg["vi"]:
        pm.q_logits.data = 0.1 * torch.randn(D, generator=g)
    if cfg["regression"] and cfg.get("pred_sig") is None:
        net.f_neg_loglike.log_std.data = 0.05 * torch.randn(cfg["O"], generator=g)
    # non-trivial BN affine + running stats so eval-mode parity means something
    for blk in model.stochstic_layers:
        bn = blk.block[2]
        bn.weight.data = 1.0 + 0.1 * torch.randn(cfg["W"], generator=g)
        bn.bias.data = 0.1 * torch.randn(cfg["W"], generator=g)
        bn.running_mean.data = 0.2 * torch.rand(cfg["W"], generator=g)
        bn.running_var.data = 0.5 + torch.rand(cfg["W"], generator=g)
    return net


def one_pass(net, cfg, x, y):
    """Forward in train mode + every intermediate the hot path defines + gradients of the loss."""
    out = {}
    net = copy.deepcopy(net)
    net.model.train()
    model, pm = net.model, net.prob_model
    D = cfg["L"] + 1
    act_vec = model.forward(x)
    act_vec.retain_grad()
    prior = model.get_w_prior_loglike(k=None)
    B = x.shape[0]
    yy = y
    rep = [D] + [1] * (len(yy.shape) - 1)
    ll = -net.f_neg_loglike(act_vec.view(D * B, -1), yy.repeat(*rep)).view(D, B)
    joint = pm.get_w_joint_loglike(prior, act_vec, yy, net.f_neg_loglike, net.N_train)
    log_marg = pm.get_marg_loglike(joint)
    log_post = pm.get_depth_log_posterior(joint, log_marg)
    out["act_vec"] = npy(act_vec)
    out["prior_ll"] = npy(prior)
    out["loglike"] = npy(ll)
    out["joint"] = npy(joint)
    out["log_marg"] = npy(log_marg)
    out["log_post"] = npy(log_post)
    if cfg["vi"]:
        ELBO = pm.estimate_ELBO(prior, act_vec, yy, net.f_neg_loglike, net.N_train, Beta=1)
        out["q"] = npy(pm.get_q_probs())
        out["KL"] = npy(pm.get_KL())
        out["E_loglike"] = npy(pm.get_E_loglike(joint))
        out["ELBO"] = npy(ELBO)
        loss = -ELBO / net.N_train
    else:
        loss = -log_marg / net.N_train
    out["loss"] = npy(loss)
    loss.backward()
    out["g/act_vec"] = npy(act_vec.grad)
    for k, p in model.named_parameters():
        out["g/" + k] = npy(p.grad)
    if cfg["vi"]:
        out["g/q_logits"] = npy(pm.q_logits.grad)
    if cfg["regression"] and cfg.get("pred_sig") is None:
        out["g/log_std"] = npy(net.f_neg_loglike.log_std.grad)
    put_sd(out, "sd_after_fwd/", {k: v for k, v in model.state_dict().items() if "running" in k or "tracked" in k})
    return out


def run_case(name, cfg, x, y, x_test=None):
    out = {"meta": json.dumps(cfg)}
    net = build(cfg)
    if not cfg["regression"]:
        y = y.long()
    out["x"], out["y"] = npy(x), npy(y)
    put_sd(out, "sd/", net.model.state_dict())
    if cfg["vi"]:
        out["q_logits0"] = npy(net.prob_model.q_logits)
    if cfg["regression"]:
        out["log_std0"] = npy(net.f_neg_loglike.log_std)
    out.update(one_pass(net, cfg, x, y))

    # three optimisation steps through the reference's own fit()
    rets = []
    posts = []
    for _ in range(3):
        rets.append(net.fit(x, y))
        posts.append(npy(net.prob_model.current_posterior))
    out["fit/ret"] = np.array(rets, dtype=np.float64)
    out["fit/posterior"] = np.stack(posts)
    put_sd(out, "fit/sd/", net.model.state_dict())
    if cfg["vi"]:
        out["fit/q_logits"] = npy(net.prob_model.q_logits)
    if cfg["regression"]:
        out["fit/log_std"] = npy(net.f_neg_loglike.log_std)

    # inference API after training
    xt = x if x_test is None else x_test
    out["x_test"] = npy(xt)
    out["eval/ret"] = np.array(net.eval(x, y), dtype=np.float64)
    out["layer_predict"] = npy(net.layer_predict(xt))
    if cfg["regression"]:
        mu, std = net.predict(xt, get_std=True)
        out["pred/mu"], out["pred/std"] = npy(mu), npy(std)
        mu, mstd = net.predict(xt, get_std=True, return_model_std=True)
        out["pred/model_std"] = npy(mstd)
        out["pred/plain"] = npy(net.predict(xt))
        if cfg["L"] >= 2 and cfg["vi"]:
            k = cfg["L"] // 2
            saved = net.prob_model.current_posterior
            qk = saved[:k + 1] / saved[:k + 1].sum()
            net.prob_model.current_posterior = qk
            mu, mstd = net.predict(xt, depth=k, get_std=True, return_model_std=True)
            out["pred/depth_k"] = np.array(k)
            out["pred/depth_k_q"] = npy(qk)
            out["pred/depth_k_mu"], out["pred/depth_k_model_std"] = npy(mu), npy(mstd)
            net.prob_model.current_posterior = saved
    else:
        out["pred/plain"] = npy(net.predict(xt))

    # dataset-level sweeps over a 3-batch loader whose length equals N_train
    if cfg.get("sweep"):
        Ns = cfg["N"]
        assert x.shape[0] >= Ns
        ds = torch.utils.data.TensorDataset(x[:Ns], y[:Ns])
        bs = (Ns + 2) // 3
        loader = torch.utils.data.DataLoader(ds, batch_size=bs, shuffle=False)
        post, lm = net.get_exact_d_posterior(loader, train_bn=False, logposterior=False)
        out["sweep/posterior"], out["sweep/log_marg"] = npy(post), np.array(lm)
        out["sweep/batch"] = np.array(bs)
        if cfg["vi"]:
            out["sweep/ELBO"] = np.array(net.get_exact_ELBO(loader, train_bn=False))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, "fit ret", rets[-1])


def edge_cases():
    """Probability-module edge semantics straight from the reference classes (no network)."""
    out = {}
    g = torch.Generator().manual_seed(7)
    # (1) regression, ragged B, clamp active (exp(log_std) < 1e-4), peaked q driving one q_d below eps
    for tag, D, B, log_std, qscale in (("clamp", 5, 37, -12.0, 1.0), ("peaked", 4, 130, 0.3, 60.0),
                                       ("one_depth", 1, 9, -0.2, 1.0), ("one_row", 3, 1, 0.1, 1.0)):
        act = torch.randn(D, B, 1, generator=g).requires_grad_(True)
        y = torch.randn(B, 1, generator=g)
        pm = quiet(depth_categorical_VI, [1.0 / D] * D, cuda=False)
        pm.q_logits.data = qscale * torch.randn(D, generator=g)
        f = homo_Gauss_mloglike(1, None)
        f.log_std.data = torch.tensor([log_std])
        N = 1000
        prior = torch.zeros(D)
        ELBO = pm.estimate_ELBO(prior, act, y, f, N, Beta=1)
        joint = depth_categorical.get_w_joint_loglike(prior, act, y, f, N)
        (-ELBO / N).backward()
        p = "reg_%s/" % tag
        out[p + "act"], out[p + "y"] = npy(act), npy(y)
        out[p + "q_logits"], out[p + "log_std"] = npy(pm.q_logits), npy(f.log_std)
        out[p + "N"] = np.array(N)
        out[p + "joint"], out[p + "ELBO"], out[p + "KL"] = npy(joint), npy(ELBO), npy(pm.get_KL())
        out[p + "q"] = npy(pm.get_q_probs())
        out[p + "g_act"], out[p + "g_q_logits"] = npy(act.grad), npy(pm.q_logits.grad)
        out[p + "g_log_std"] = npy(f.log_std.grad)
        lm = pm.get_marg_loglike(joint)
        out[p + "log_marg"] = npy(lm)
        out[p + "log_post"] = npy(pm.get_depth_log_posterior(joint, lm))
        mu, std = depth_categorical.marginalise_d_predict(act.data, pm.get_q_probs(), get_std=True)
        out[p + "mu"], out[p + "std"] = npy(mu), npy(std)
    # (2) classification, several class counts incl. non power of two and > 64
    for C, D, B in ((10, 12, 50), (3, 2, 7), (100, 3, 33), (1000, 2, 5)):
        act = (3.0 * torch.randn(D, B, C, generator=g)).requires_grad_(True)
        y = torch.randint(0, C, (B,), generator=g)
        pm = quiet(depth_categorical_VI, [1.0 / D] * D, cuda=False)
        pm.q_logits.data = torch.randn(D, generator=g)
        f = torch.nn.CrossEntropyLoss(reduction="none")
        N = 5000
        prior = torch.zeros(D)
        ELBO = pm.estimate_ELBO(prior, act, y, f, N, Beta=1)
        joint = depth_categorical.get_w_joint_loglike(prior, act, y, f, N)
        (-ELBO / N).backward()
        p = "cls_%d/" % C
        out[p + "act"], out[p + "y"] = npy(act), npy(y)
        out[p + "q_logits"] = npy(pm.q_logits)
        out[p + "N"] = np.array(N)
        out[p + "joint"], out[p + "ELBO"] = npy(joint), npy(ELBO)
        out[p + "g_act"], out[p + "g_q_logits"] = npy(act.grad), npy(pm.q_logits.grad)
        act2 = act.detach().clone().requires_grad_(True)
        joint2 = depth_categorical.get_w_joint_loglike(prior, act2, y, f, N)
        lm = pm.get_marg_loglike(joint2)
        (-lm / N).backward()
        out[p + "log_marg"] = npy(lm)
        out[p + "gm_act"] = npy(act2.grad)
        out[p + "probs"] = npy(depth_categorical.marginalise_d_predict(act.data, pm.get_q_probs(), softmax=True))
    # (3) O=2 regression must raise in the reference
    try:
        act = torch.randn(3, 5, 2)
        depth_categorical.get_w_joint_loglike(torch.zeros(3), act, torch.randn(5, 2), homo_Gauss_mloglike(2), 10)
        out["o2_raises"] = np.array(0)
    except RuntimeError:
        out["o2_raises"] = np.array(1)
    np.savez_compressed(os.path.join(HERE, "edge.npz"), **out)
    print("wrote edge")


def head_case():
    """Reduced image-DUN head: shared Linear over per-depth pooled features + CE ELBO (C4/C5 shape family)."""
    g = torch.Generator().manual_seed(11)
    D, B, K, C = 6, 40, 512, 10
    feat = torch.randn(D, B, K, generator=g).relu().requires_grad_(True)
    lin = torch.nn.Linear(K, C)
    y = torch.randint(0, C, (B,), generator=g)
    pm = quiet(depth_categorical_VI, [1.0 / D] * D, cuda=False)
    pm.q_logits.data = 0.3 * torch.randn(D, generator=g)
    N = 60000
    act = lin(feat.view(D * B, K)).view(D, B, C)
    ELBO = pm.estimate_ELBO(torch.zeros(D), act, y, torch.nn.CrossEntropyLoss(reduction="none"), N, Beta=1)
    (-ELBO / N).backward()
    out = dict(feat=npy(feat), w=npy(lin.weight), b=npy(lin.bias), y=npy(y), q_logits=npy(pm.q_logits),
               N=np.array(N), act=npy(act), ELBO=npy(ELBO), g_feat=npy(feat.grad), g_w=npy(lin.weight.grad),
               g_b=npy(lin.bias.grad), g_q_logits=npy(pm.q_logits.grad))
    np.savez_compressed(os.path.join(HERE, "head_c10.npz"), **out)
    print("wrote head_c10")


def image_case():
    """Reduced image DUN (C4/C5 family): ResNet-18 DUN on 16x16 inputs.  The 11 M weights are not stored: the
    fixture pins the torch seed, and dun_amd's constructor draws its initial weights in the reference's order."""
    with contextlib.redirect_stdout(io.StringIO()):
        from src.DUN.stochastic_img_resnets import resnet18
    for tag, kw, shape in (("img_r18", dict(start_depth=1, end_depth=7, num_classes=10, zero_init_residual=True,
                                            initial_conv="1x3", concat_pool=False, input_chanels=3, p_drop=0),
                            (6, 3, 16, 16)),
                           ("img_r18_mnist", dict(start_depth=2, end_depth=4, num_classes=10, zero_init_residual=False,
                                                  initial_conv="3x3", concat_pool=True, input_chanels=1, p_drop=0),
                            (5, 1, 28, 28))):
        torch.manual_seed(0)
        model = resnet18(arch_uncert=True, **kw)
        D = model.n_layers
        g = torch.Generator().manual_seed(21)
        x = torch.randn(*shape, generator=g)
        y = torch.randint(0, 10, (shape[0],), generator=g)
        pm = quiet(depth_categorical_VI, [1.0 / D] * D, cuda=False)
        pm.q_logits.data = 0.2 * torch.randn(D, generator=g)
        net = quiet(DUN_VI, model, pm, 60000, lr=0.1, momentum=0.9, weight_decay=1e-4, cuda=False, regression=False)
        out = {"meta": json.dumps(dict(kw=kw, shape=shape, D=D, N=60000, lr=0.1, momentum=0.9, wd=1e-4)),
               "x": npy(x), "y": npy(y), "q_logits0": npy(pm.q_logits)}
        out["n_params"] = np.array(sum(p.numel() for p in model.parameters()))
        out["keys"] = np.array(sorted(model.state_dict().keys()))
        # weight fingerprints (seed-matched construction check)
        sd = model.state_dict()
        for k in ("conv1.0.weight", "layer_list.0.conv1.weight", "layer_list.7.conv2.weight", "adapt0.0.weight",
                  "adapt2.0.weight", "output_block.weight", "output_block.bias"):
            out["w/" + k] = npy(sd[k]).ravel()[:64]
        m2 = copy.deepcopy(model)
        m2.train()
        act = m2.forward(x)
        act.retain_grad()
        ELBO = pm.estimate_ELBO(m2.get_w_prior_loglike(), act, y, net.f_neg_loglike, net.N_train, Beta=1)
        (-ELBO / net.N_train).backward()
        out["act_vec"], out["ELBO"], out["g/act_vec"] = npy(act), npy(ELBO), npy(act.grad)
        out["g/output_block.weight"] = npy(m2.output_block.weight.grad)
        out["g/output_block.bias"] = npy(m2.output_block.bias.grad)
        out["g/conv1.0.weight"] = npy(m2.conv1[0].weight.grad)
        out["g/q_logits"] = npy(pm.q_logits.grad)
        pm.q_logits.grad = None
        m2.eval()
        with torch.no_grad():
            out["act_eval"] = npy(m2.forward(x))
        rets = [net.fit(x, y) for _ in range(2)]
        out["fit/ret"] = np.array(rets, dtype=np.float64)
        out["fit/q_logits"] = npy(pm.q_logits)
        out["eval/ret"] = np.array(net.eval(x, y), dtype=np.float64)
        out["pred"] = npy(net.predict(x))
        peaked = torch.zeros(D)
        peaked[1], peaked[D - 1] = 0.7, 0.3
        net.prob_model.current_posterior = peaked
        out["fast_pred"] = npy(quiet(net.fast_predict, x))
        out["fast_pred_q"] = npy(peaked)
        np.savez_compressed(os.path.join(HERE, tag + ".npz"), **out)
        print("wrote", tag, "fit ret", rets[-1], "D", D)


def image50_case():
    """The C4 / C5 constructors themselves: ResNet-50 DUN, start_depth=1, end_depth=13 (D = 12 exits), 10 classes,
    1x3 stem (experiments/image/train_DUN.py:45-48, 103-139), at batch 4 on MNIST- and CIFAR-shaped inputs.  The
    26 M weights are not stored: the torch seed pins them, fingerprints of a few tensors check the draw order."""
    with contextlib.redirect_stdout(io.StringIO()):
        from src.DUN.stochastic_img_resnets import resnet50
    for tag, chans, side in (("img_r50_mnist", 1, 28), ("img_r50_cifar", 3, 32)):
        kw = dict(start_depth=1, end_depth=13, num_classes=10, zero_init_residual=True, initial_conv="1x3",
                  concat_pool=False, input_chanels=chans, p_drop=0)
        torch.manual_seed(0)
        model = resnet50(arch_uncert=True, **kw)
        D = model.n_layers
        g = torch.Generator().manual_seed(22)
        x = torch.randn(4, chans, side, side, generator=g)
        y = torch.randint(0, 10, (4,), generator=g)
        pm = quiet(depth_categorical_VI, [1.0 / D] * D, cuda=False)
        pm.q_logits.data = 0.2 * torch.randn(D, generator=g)
        N = 60000 if chans == 1 else 50000
        net = quiet(DUN_VI, model, pm, N, lr=0.1, momentum=0.9, weight_decay=1e-4, cuda=False, regression=False)
        out = {"meta": json.dumps(dict(kw=kw, shape=[4, chans, side, side], D=D, N=N, lr=0.1, momentum=0.9, wd=1e-4,
                                       channel_list=[int(c) for c in model.channel_list])),
               "x": npy(x), "y": npy(y), "q_logits0": npy(pm.q_logits)}
        out["n_params"] = np.array(sum(p.numel() for p in model.parameters()))
        out["keys"] = np.array(sorted(model.state_dict().keys()))
        sd = model.state_dict()
        for k in ("conv1.0.weight", "layer_list.0.conv1.weight", "layer_list.8.conv3.weight", "layer_list.15.conv2.weight",
                  "adapt0.0.weight", "adapt1.0.weight", "adapt2.0.weight", "output_block.weight", "output_block.bias"):
            out["w/" + k] = npy(sd[k]).ravel()[:64]
        m2 = copy.deepcopy(model)
        m2.train()
        act = m2.forward(x)
        act.retain_grad()
        ELBO = pm.estimate_ELBO(m2.get_w_prior_loglike(), act, y, net.f_neg_loglike, net.N_train, Beta=1)
        (-ELBO / net.N_train).backward()
        out["act_vec"], out["ELBO"], out["g/act_vec"] = npy(act), npy(ELBO), npy(act.grad)
        out["g/output_block.weight"] = npy(m2.output_block.weight.grad)
        out["g/output_block.bias"] = npy(m2.output_block.bias.grad)
        out["g/q_logits"] = npy(pm.q_logits.grad)
        pm.q_logits.grad = None
        m2.eval()
        with torch.no_grad():
            out["act_eval"] = npy(m2.forward(x))
        out["fit/ret"] = np.array([net.fit(x, y)], dtype=np.float64)
        np.savez_compressed(os.path.join(HERE, tag + ".npz"), **out)
        print("wrote", tag, "D", D, "params", int(out["n_params"]), "fit", out["fit/ret"])


def image_exits_case():
    """Everything BEHIND the convolutional trunk of the C5 constructor (ResNet-50 DUN, D = 12 exits, CIFAR-shaped input,
    batch 4), pinned so that it can be checked without the trunk: the pooled [D,B,2048] features that reach the shared
    output_block at every exit (captured by a hook on output_block), and what the reference makes of them --
    act_vec, ELBO and its gradients w.r.t. act_vec / the features / output_block / q_logits in train mode, and in eval
    mode act_vec, predict, and fast_forward_impl / fast_predict (src/DUN/stochastic_img_resnets.py:288-302,
    src/DUN/training_wrappers.py:160-169) under a posterior with one exit below min_prob = 1e-2 (5e-3), one just
    above it (2e-2) and the rest of the mass on two exits."""
    with contextlib.redirect_stdout(io.StringIO()):
        from src.DUN.stochastic_img_resnets import resnet50
    kw = dict(start_depth=1, end_depth=13, num_classes=10, zero_init_residual=False, initial_conv="1x3",
              concat_pool=False, input_chanels=3, p_drop=0)
    torch.manual_seed(0)
    model = resnet50(arch_uncert=True, **kw)
    D = model.n_layers
    g = torch.Generator().manual_seed(23)
    x = torch.randn(4, 3, 32, 32, generator=g)
    y = torch.randint(0, 10, (4,), generator=g)
    pm = quiet(depth_categorical_VI, [1.0 / D] * D, cuda=False)
    pm.q_logits.data = 0.3 * torch.randn(D, generator=g)
    with torch.no_grad():   # a head that is not at its near-zero initial scale: logits of order 1
        model.output_block.weight.mul_(3.0)
        model.output_block.bias.copy_(0.1 * torch.randn(10, generator=g))
    N = 50000
    net = quiet(DUN_VI, model, pm, N, lr=0.1, momentum=0.9, weight_decay=1e-4, cuda=False, regression=False)
    out = {"meta": json.dumps(dict(kw=kw, shape=[4, 3, 32, 32], D=D, N=N, lr=0.1, momentum=0.9, wd=1e-4)),
           "x": npy(x), "y": npy(y), "q_logits0": npy(pm.q_logits),
           "output_block.weight": npy(model.output_block.weight), "output_block.bias": npy(model.output_block.bias)}
    seen = []
    hook = model.output_block.register_forward_pre_hook(lambda mod, inp: seen.append(inp[0]))
    model.train()
    act = model.forward(x)
    act.retain_grad()
    for f in seen:
        f.retain_grad()
    ELBO = pm.estimate_ELBO(model.get_w_prior_loglike(), act, y, net.f_neg_loglike, N, Beta=1)
    (-ELBO / N).backward()
    assert len(seen) == D
    out["train/feat"] = np.stack([npy(f) for f in seen])
    out["train/g_feat"] = np.stack([npy(f.grad) for f in seen])
    out["train/act_vec"], out["train/ELBO"], out["train/g_act_vec"] = npy(act), npy(ELBO), npy(act.grad)
    out["train/g_output_block.weight"] = npy(model.output_block.weight.grad)
    out["train/g_output_block.bias"] = npy(model.output_block.bias.grad)
    out["train/g_q_logits"] = npy(pm.q_logits.grad)
    pm.q_logits.grad = None
    model.zero_grad()
    del seen[:]
    model.eval()
    with torch.no_grad():
        out["eval/act_vec"] = npy(model.forward(x))
        out["eval/feat"] = np.stack([npy(f) for f in seen])
        del seen[:]
        pm.current_posterior = pm.get_q_probs().data.clone()   # (what fit leaves behind, src/DUN/training_wrappers.py:226)
        out["posterior"] = npy(pm.current_posterior)
        out["eval/ret"] = np.array(net.eval(x, y), dtype=np.float64)
        del seen[:]
        out["pred"] = npy(net.predict(x))
        del seen[:]
        q = torch.zeros(D)
        q[2], q[5], q[7], q[11] = 5e-3, 2e-2, 0.6, 0.375
        net.prob_model.current_posterior = q
        out["fast/q"] = npy(q)
        out["fast/act_vec"] = npy(quiet(model.fast_forward_impl, x, q, min_prob=1e-2))
        del seen[:]
        out["fast/pred"] = npy(quiet(net.fast_predict, x))
    hook.remove()
    zero_rows = [d for d in range(D) if not out["fast/act_vec"][d].any()]
    assert zero_rows == [0, 1, 2, 3, 4, 6, 8, 9, 10], zero_rows
    np.savez_compressed(os.path.join(HERE, "img_r50_cifar_exits.npz"), **out)
    print("wrote img_r50_cifar_exits", "ELBO", float(ELBO), "zero rows", zero_rows, "|act|max", float(np.abs(out["eval/act_vec"]).max()))


def uci_case():
    """A tiny synthetic table in the reference's UCI on-disk layout, read by the reference's load_gap_UCI
    (src/datasets/UCI_gap_loader.py:12-60); raw files and outputs go into uci_small.npz."""
    import tempfile
    from src.datasets.UCI_gap_loader import load_gap_UCI
    rs = np.random.RandomState(42)
    n, nf = 40, 5
    table = np.round(rs.randn(n, nf + 1) * np.array([1, 10, 0.1, 3, 1, 2]) + np.array([0, 5, 1, -2, 0, 1]), 4)
    table[:, 3] = 7.0  # a constant feature: std < 1e-10 -> 1
    feat, targ = np.array([0, 1, 3, 4, 5]), np.array(2)
    out = {"table": table, "feature_idx": feat, "target_idx": targ}
    with tempfile.TemporaryDirectory() as tmp:
        for kind in ("standard", "gap"):
            d = os.path.join(tmp, "UCI_for_sharing", kind, "toy", "data")
            os.makedirs(d)
            if kind == "standard":
                np.savetxt(os.path.join(d, "data.txt"), table)
                np.savetxt(os.path.join(d, "index_features.txt"), feat, fmt="%d")
                np.savetxt(os.path.join(d, "index_target.txt"), targ.reshape(1), fmt="%d")
            for i in range(3):
                perm = rs.permutation(n)
                k = 30 if kind == "standard" else 26
                out["%s_train_%d" % (kind, i)], out["%s_test_%d" % (kind, i)] = perm[:k], perm[k:]
                np.savetxt(os.path.join(d, "index_train_%d.txt" % i), perm[:k], fmt="%d")
                np.savetxt(os.path.join(d, "index_test_%d.txt" % i), perm[k:], fmt="%d")
        names = ["X_train", "X_test", "x_means", "x_stds", "y_train", "y_test", "y_means", "y_stds"]
        for gap in (True, False):
            for sp in (0, 2):
                res = load_gap_UCI(tmp, "toy", n_split=sp, gap=gap)
                for nm, v in zip(names, res):
                    out["out/%s/%d/%s" % ("gap" if gap else "standard", sp, nm)] = np.asarray(v)
    np.savez_compressed(os.path.join(HERE, "uci_small.npz"), **out)
    print("wrote uci_small")


def label_cases():
    """Label semantics of the reference's classification losses (src/DUN/training_wrappers.py:29-30, :85-87):
    ignore_index rows through CrossEntropyLoss(reduction='none') in the ELBO and through NLLLoss / pred.ne(y) in the
    metrics; a label outside [0, C) raises."""
    out = {}
    g = torch.Generator().manual_seed(31)
    f = torch.nn.CrossEntropyLoss(reduction="none")
    f_test = torch.nn.NLLLoss(reduction="none")
    for C, D, B in ((10, 4, 64), (100, 3, 36), (3, 2, 7)):
        act = (2.0 * torch.randn(D, B, C, generator=g)).requires_grad_(True)
        y = torch.randint(0, C, (B,), generator=g)
        y[torch.randperm(B, generator=g)[:max(2, B // 5)]] = -100
        pm = quiet(depth_categorical_VI, [1.0 / D] * D, cuda=False)
        pm.q_logits.data = 0.5 * torch.randn(D, generator=g)
        N = 777
        prior = torch.zeros(D)
        ELBO = pm.estimate_ELBO(prior, act, y, f, N, Beta=1)
        joint = depth_categorical.get_w_joint_loglike(prior, act, y, f, N)
        (-ELBO / N).backward()
        p = "c%d/" % C
        out[p + "act"], out[p + "y"], out[p + "q_logits"], out[p + "N"] = npy(act), npy(y), npy(pm.q_logits), np.array(N)
        out[p + "loglike"] = npy(-f(act.view(D * B, C), y.repeat(D)).view(D, B))
        out[p + "joint"], out[p + "ELBO"] = npy(joint), npy(ELBO)
        out[p + "g_act"], out[p + "g_q_logits"] = npy(act.grad), npy(pm.q_logits.grad)
        probs = depth_categorical.marginalise_d_predict(act.data, pm.get_q_probs(), softmax=True)
        out[p + "nll"] = npy(f_test(torch.log(probs), y).mean(dim=0))
        out[p + "err"] = np.array(probs.max(dim=1)[1].ne(y).sum().item() / y.shape[0])
        for tag, bad in (("hi", C), ("neg", -1)):
            yb = y.clone()
            yb[0] = bad
            try:
                f(act.detach().view(D * B, C), yb.repeat(D))
                out[p + "raises_" + tag] = np.array(0)
            except IndexError:
                out[p + "raises_" + tag] = np.array(1)
    np.savez_compressed(os.path.join(HERE, "labels.npz"), **out)
    print("wrote labels")


def prior_cases():
    """Weight-prior path: diag_w_Gauss_loglike plain and with BMA_prior, VI and exact-marginal."""
    base = dict(seed=0, O=1, vi=True, regression=True, wd=0.0)
    wp = dict(mu=0.0, sig2=0.3, mult=1.0)
    x, y = synthetic_regression(160, 6, 9)
    run_case("prior_gauss", dict(base, din=6, W=48, L=3, N=160, lr=1e-2, momentum=0.5, w_prior=wp, sweep=True), x, y)
    run_case("prior_bma", dict(base, din=6, W=48, L=3, N=160, lr=1e-2, momentum=0.5, bma=True,
                               w_prior=dict(mu=0.05, sig2=0.5, mult=0.5), sweep=True), x, y)
    # (N_train 16 against a batch of 160 keeps the exact posterior off one-hot: with a one-hot posterior the predictive
    #  variance lands on either side of 0 and torch >= 2 refuses the NaN scale inside the reference's Normal)
    run_case("prior_mll", dict(base, vi=False, din=6, W=48, L=3, N=16, lr=1e-2, momentum=0.5, w_prior=wp), x, y)
    g = torch.Generator().manual_seed(10)
    xc = torch.randn(96, 12, generator=g)
    yc = torch.randint(0, 5, (96,), generator=g)
    run_case("prior_cls", dict(base, regression=False, din=12, O=5, W=32, L=2, N=96, lr=0.05, momentum=0.9, bma=True,
                               w_prior=wp), xc, yc)


def checkpoint_case():
    """Checkpoint ingestion: a reference DUN_VI with a MultiStepLR schedule is trained for two steps
    and one epoch tick, its state written in the tensors-only layout (dun_amd.checkpoint.net_state reads the
    reference's own objects), and training continued -- a step, an epoch tick that crosses a milestone, another step.
    The continuation is what a wrapper restored from the file must reproduce (momentum and schedule included)."""
    from dun_amd.checkpoint import net_state
    cfg = dict(din=5, O=1, W=16, L=3, N=96, lr=5e-2, momentum=0.9, wd=1e-4, regression=True, vi=True, seed=11)
    x, y = synthetic_regression(64, cfg["din"], 21)
    net = build(cfg)
    quiet(net.make_scheduler, gamma=0.1, milestones=[2, 4])
    out = {"meta": json.dumps(cfg), "x": npy(x), "y": npy(y)}
    for _ in range(2):
        net.fit(x, y)
    net.update_lr()
    torch.save(net_state(net, best_err=0.25), os.path.join(HERE, "ckpt_ref_vi.pt"))
    rets = [net.fit(x, y)]
    net.update_lr()  # epoch 2: first milestone, lr x 0.1
    out["lr_after"] = np.array(net.optimizer.param_groups[0]["lr"])
    rets.append(net.fit(x, y))
    out["fit/ret"] = np.array(rets, dtype=np.float64)
    put_sd(out, "fit/sd/", net.model.state_dict())
    out["fit/q_logits"] = npy(net.prob_model.q_logits)
    out["fit/log_std"] = npy(net.f_neg_loglike.log_std)
    np.savez_compressed(os.path.join(HERE, "ckpt_ref_vi.npz"), **out)
    print("wrote ckpt_ref_vi", rets, float(out["lr_after"]))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "prior":
        prior_cases()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "image50":
        image50_case()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "labels":
        label_cases()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "image":
        image_case()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "exits":
        image_exits_case()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "uci":
        uci_case()
        return
    base = dict(seed=0, O=1, vi=True, regression=True, wd=0.0)
    X, Y = load_wiggle()
    run_case("c1_wiggle", dict(base, din=1, W=100, L=10, N=300, lr=1e-3, momentum=0.9, wd=1e-4, sweep=True),
             torch.from_numpy(X), torch.from_numpy(Y))
    x, y = synthetic_regression(512, 13, 1)
    run_case("c2_boston", dict(base, din=13, W=100, L=10, N=455, lr=1e-2, momentum=0.5), x, y)
    x, y = synthetic_regression(512, 9, 2)
    run_case("c2_protein", dict(base, din=9, W=100, L=10, N=41157, lr=1e-2, momentum=0.5), x, y)
    x, y = synthetic_regression(1024, 64, 3)
    run_case("c3_reduced", dict(base, din=64, W=256, L=3, N=1 << 20, lr=1e-2, momentum=0.5), x, y)
    x, y = synthetic_regression(200, 5, 4)
    run_case("mll_small", dict(base, vi=False, din=5, W=48, L=4, N=200, lr=1e-2, momentum=0.5, sweep=True), x, y)
    x, y = synthetic_regression(96, 7, 5)
    run_case("mlp_nores", dict(base, mlp=True, din=7, W=40, L=3, N=96, lr=1e-2, momentum=0.9, wd=1e-3), x, y)
    x, y = synthetic_regression(64, 3, 6)
    run_case("fixed_sig", dict(base, din=3, W=32, L=2, N=64, lr=1e-2, momentum=0.5, pred_sig=0.5), x, y)
    g = torch.Generator().manual_seed(8)
    xc = torch.randn(256, 20, generator=g)
    yc = torch.randint(0, 10, (256,), generator=g)
    run_case("cls_fc", dict(base, regression=False, din=20, O=10, W=64, L=4, N=256, lr=0.05, momentum=0.9,
                            wd=1e-4, sweep=True), xc, yc)
    run_case("cls_fc_mll", dict(base, regression=False, vi=False, din=20, O=10, W=64, L=4, N=256, lr=0.05,
                                momentum=0.9), xc, yc)
    prior_cases()
    checkpoint_case()
    label_cases()
    edge_cases()
    head_case()
    image_case()
    image50_case()
    image_exits_case()
    uci_case()


if __name__ == "__main__":
    main()
