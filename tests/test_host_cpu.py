"""CPU-side checks: the C-ABI library loads and exports what include/dun_hip.h declares; the host logic above the
ABI (wrappers, optimizer, autograd plumbing) reproduces the reference's goldens when driven by the CPU stand-in
backend; and the product backend refuses CPU tensors instead of falling back."""
import contextlib
import ctypes
import io
import os
import re

import numpy as np
import pytest
import torch

from tests.util import Case, model_prior_kwargs, rel_err

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def test_library_exports_every_declared_symbol():
    from dun_amd import _lib
    from dun_amd.build import build_library
    lib = ctypes.CDLL(build_library())
    header = open(os.path.join(ROOT, "include", "dun_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(dun_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), "libdun_hip.so does not export %s" % name
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    assert lib.dun_abi_version() >= 2


def test_no_cpu_fallback_in_product_backend():
    from dun_amd import kernels
    kernels.set_backend(None)
    be = kernels.backend()
    assert be.name == "hip"
    with pytest.raises(RuntimeError, match="GPU"):
        be.ll_pass(torch.zeros(2, 3, 1), torch.zeros(3, 1), True, torch.zeros(1), None, False, 1.0)
    from dun_amd.stochastic_fc_models import arq_uncert_fc_resnet
    m = arq_uncert_fc_resnet(3, 1, 8, 2)
    with pytest.raises(RuntimeError):
        m.forward(torch.zeros(4, 3))


def test_argument_validation_without_gpu():
    """Host-side validation of the C entries rejects bad arguments before anything is launched."""
    from dun_amd import _lib as L
    lib = L.lib()
    assert lib.dun_ll_gauss(None, None, None, None, 0, 3, 5, 1.0, None, None, None, None) == 1
    assert b"null" in lib.dun_last_error()
    assert lib.dun_sgd_step(None, None, None, 10, 0.1, 0.9, 0.0, None) == 1
    d = L.GemmDesc()
    assert lib.dun_gemm_f32(ctypes.byref(d), None) == 1
    assert lib.dun_ll_num_partials(65536, 1) == 16 and lib.dun_ll_num_partials(0, 1) == 0
    assert lib.dun_fc_workspace_floats(512, 13, 100, 1, 10) > 0


@pytest.fixture
def cpu_backend():
    from dun_amd import kernels
    from tests.cpu_backend import CpuBackend
    kernels.set_backend(CpuBackend())
    yield
    kernels.set_backend(None)


def build_cpu_net(c):
    from dun_amd.probability import depth_categorical, depth_categorical_VI
    from dun_amd.stochastic_fc_models import arq_uncert_fc_MLP, arq_uncert_fc_resnet
    from dun_amd.training_wrappers import DUN, DUN_VI
    cfg = c.cfg
    ctor = arq_uncert_fc_MLP if cfg.get("mlp") else arq_uncert_fc_resnet
    model = ctor(cfg["din"], cfg["O"], cfg["W"], cfg["L"], **model_prior_kwargs(c))
    model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in c.sub("sd/").items()})
    pm = (depth_categorical_VI if cfg["vi"] else depth_categorical)(c.prior(), cuda=False)
    net = quiet(DUN_VI if cfg["vi"] else DUN, model, pm, cfg["N"], lr=cfg["lr"], momentum=cfg["momentum"],
                weight_decay=cfg["wd"], cuda=False, regression=cfg["regression"], pred_sig=cfg.get("pred_sig"))
    if cfg["vi"]:
        pm.q_logits.data.copy_(c.t("q_logits0"))
    if cfg["regression"]:
        net.f_neg_loglike.log_std.data.copy_(c.t("log_std0"))
    return net


@pytest.mark.parametrize("name", ["c2_boston", "mlp_nores", "fixed_sig", "cls_fc", "cls_fc_mll", "mll_small",
                                  "prior_gauss", "prior_bma", "prior_mll", "prior_cls"])
def test_host_logic_reproduces_goldens(cpu_backend, name):
    c = Case(name)
    cfg = c.cfg
    net = build_cpu_net(c)
    x, y = c.t("x"), c.y()
    for step in range(3):
        ret = np.array(net.fit(x, y))
        want = c["fit/ret"][step]
        keep = [0, 2] if np.isnan(ret[1]) else [0, 1, 2]
        assert np.allclose(ret[keep], want[keep], rtol=2e-5 if step == 0 else 2e-3, atol=1e-6), (step, ret, want)
    if cfg["vi"]:
        assert rel_err(net.prob_model.q_logits.data, c["fit/q_logits"]) < 2e-3
    sd = net.model.state_dict()
    assert rel_err(sd["input_layer.weight"], c["fit/sd/input_layer.weight"]) < 2e-3
    assert int(sd["stochstic_layers.0.block.2.num_batches_tracked"]) == 3
    ev = np.array(net.eval(x, y))
    keep = [1] if np.isnan(ev[0]) else [0, 1]
    assert np.allclose(ev[keep], c["eval/ret"][keep], rtol=5e-3)
    assert rel_err(net.predict(c.t("x_test")), c["pred/plain"]) < 5e-3
    if cfg.get("sweep"):
        bs, N = int(c["sweep/batch"]), cfg["N"]
        loader = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(x[:N], y[:N]), batch_size=bs)
        post, lm = net.get_exact_d_posterior(loader)
        assert abs(lm - float(c["sweep/log_marg"])) < 5e-3 * abs(lm)
        if cfg["vi"]:
            assert abs(net.get_exact_ELBO(loader) - float(c["sweep/ELBO"])) < 5e-3 * abs(float(c["sweep/ELBO"]))


@pytest.mark.parametrize("name", ["prior_gauss", "prior_bma", "prior_mll", "prior_cls"])
def test_weight_prior_value_and_gradient(cpu_backend, name):
    """Row a5: get_w_prior_loglike returns the reference's scalar (plain and BMA_prior), the product's
    diag_w_Gauss_loglike is the reference's density, and one backward leaves the reference's gradients (data term from
    the trunk backward + prior term through autograd) in the flat gradient buffer."""
    from dun_amd.probability import diag_w_Gauss_loglike
    from tests.util import stacked_grads_from_reference
    c = Case(name)
    cfg = c.cfg
    net = build_cpu_net(c)
    prior = net.model.get_w_prior_loglike(k=None)
    assert prior.dim() == 0 and prior.requires_grad
    assert rel_err(prior.detach(), c["prior_ll"]) < 2e-5
    wp = cfg["w_prior"]
    net.model.w_prior = [diag_w_Gauss_loglike(wp["mu"], wp["sig2"], wp["mult"])] * (cfg["L"] + 1)
    assert rel_err(net.model.get_w_prior_loglike().detach(), c["prior_ll"]) < 2e-5
    x, y = c.t("x"), c.y()
    net.model.train()
    act = net.model.forward(x)
    prior = net.model.get_w_prior_loglike(k=None)
    out = net.prob_model.fused_objective(prior, act, y, net.f_neg_loglike, cfg["N"])
    assert rel_err(out.joint, c["joint"]) < 2e-5
    assert rel_err(out.objective, c["ELBO"] if cfg["vi"] else c["log_marg"]) < 2e-5
    net._backward(act, prior, out)
    for k, v in stacked_grads_from_reference(c).items():
        assert rel_err(net.model.stacked_grad[k], v) < 5e-4, k
    # the prior term is not negligible in these cases: without it the weight gradient would be off by far more
    net2 = build_cpu_net(c)
    net2.model.train()
    act2 = net2.model.forward(x)
    out2 = net2.prob_model.fused_objective(prior.detach(), act2, y, net2.f_neg_loglike, cfg["N"])
    net2._backward(act2, prior.detach(), out2)
    assert rel_err(net2.model.stacked_grad["w"], stacked_grads_from_reference(c)["w"]) > 1e-2


def check_label_semantics_through_the_wrapper(net_builder):
    """fit / eval with an ignored (-100) label run and agree with the oracle; any other label outside [0, C) raises
    IndexError like nn.CrossEntropyLoss does on CPU -- from the step's one host copy."""
    c = Case("cls_fc")
    x, y = c.t("x"), c.y().clone()
    y[3] = y[17] = -100
    net = net_builder(c)
    onet = c.oracle_net()
    for step in range(2):
        got, want = net.fit(x, y), onet.fit(x, y)
        assert np.allclose(got, want, rtol=2e-5 if step == 0 else 5e-3), (got, want)
    onet.current_posterior = onet.current_posterior.detach()
    assert np.allclose(net.eval(x, y), onet.evaluate(x, y), rtol=5e-3)
    def snapshot(n):
        st = {"flat": n.model.flat_params.detach().clone()}
        for i, p in enumerate(n.prob_model.parameters()):
            st["prob%d" % i] = p.detach().clone()
        for i, (k, v) in enumerate(sorted(n.optimizer.state.items(), key=lambda kv: id(kv[0]))):
            for name, t in v.items():
                if torch.is_tensor(t):
                    st["opt%d/%s" % (i, name)] = t.detach().clone()
        return st

    for name in ("cls_fc", "cls_fc_mll"):          # q_logits is a parameter in the first, there is none in the second
        c2 = Case(name)
        for bad in (10, -1):
            yb = c2.y().clone()
            yb[5] = bad
            nb = net_builder(c2)
            nb.fit(x, c2.y())                        # a valid step first: momentum buffers exist and are not zero
            before = snapshot(nb)
            assert len(before) >= 2
            with pytest.raises(IndexError):
                nb.fit(x, yb)
            after = snapshot(nb)
            # the reference raises inside the loss, BEFORE its update: nothing the optimiser owns may have moved, and
            # in particular the NaN objective of such a batch must not have reached q_logits or a momentum buffer
            assert sorted(after) == sorted(before)
            for k in before:
                assert bool(torch.isfinite(after[k]).all()), k
                assert torch.equal(after[k], before[k]), k
            with pytest.raises(IndexError):
                nb.fit(x, yb)                        # (and again: on the GPU this is a replay of the captured step)
            assert all(torch.equal(v, before[k]) for k, v in snapshot(nb).items())
            ret = nb.fit(x, c2.y())                  # training goes on from where it was
            assert np.all(np.isfinite(np.array(ret, dtype=np.float64)[[0, 2]]))
            assert not torch.equal(nb.model.flat_params, before["flat"])
    yb = c.y().clone()
    yb[5] = 10
    with pytest.raises(IndexError):
        net.eval(x, yb)


def test_label_semantics_through_the_wrapper(cpu_backend):
    check_label_semantics_through_the_wrapper(build_cpu_net)


def test_backward_after_the_buffers_were_reused_raises(cpu_backend):
    """One buffer set per (batch, depth, mode): a second train-mode forward before the first one's backward must not
    silently produce gradients from the wrong activations."""
    c = Case("c2_boston")
    net = build_cpu_net(c)
    x, y = c.t("x"), c.y()
    net.model.train()
    act1 = net.model.forward(x)
    out1 = net.prob_model.fused_objective(None, act1, y, net.f_neg_loglike, c.cfg["N"])
    with torch.no_grad():
        net.model.forward(x)          # e.g. get_exact_ELBO(train_bn=True) between forward and backward
    with pytest.raises(RuntimeError, match="reused"):
        net._backward(act1, None, out1)
    act2 = net.model.forward(x)       # the normal order still works
    out2 = net.prob_model.fused_objective(None, act2, y, net.f_neg_loglike, c.cfg["N"])
    net._backward(act2, None, out2)


def test_q_warmup_and_scheduler(cpu_backend):
    """Callers freeze q_logits for the first epochs (src/DUN/train_fc.py:51-57): a frozen q must not move."""
    c = Case("c2_boston")
    net = build_cpu_net(c)
    q0 = net.prob_model.q_logits.data.clone()
    net.prob_model.q_logits.requires_grad = False
    net.fit(c.t("x"), c.y())
    assert torch.equal(net.prob_model.q_logits.data, q0)
    net.prob_model.q_logits.requires_grad = True
    net.fit(c.t("x"), c.y())
    assert not torch.equal(net.prob_model.q_logits.data, q0)
    net.make_scheduler(gamma=0.1, milestones=[1])
    net.update_lr()
    assert abs(net.optimizer.param_groups[0]["lr"] - 0.1 * c.cfg["lr"]) < 1e-12 and net.epoch == 1


def test_train_fc_loop_plumbing(cpu_backend, tmp_path):
    """Config-1 plumbing on CPU (stand-in backend): the epoch loop trains, tracks, checkpoints and stops early."""
    from dun_amd.datasets import Datafeed, load_wiggle
    from dun_amd.probability import depth_categorical_VI
    from dun_amd.stochastic_fc_models import arq_uncert_fc_resnet
    from dun_amd.train_fc import train_fc_DUN
    from dun_amd.training_wrappers import DUN_VI
    from oracle.dun_oracle import wiggle_data
    X, Y = load_wiggle()
    Xo, Yo = wiggle_data()
    assert np.array_equal(X, Xo) and np.array_equal(Y, Yo)
    ds = Datafeed(torch.Tensor(X), torch.Tensor(Y))
    train = torch.utils.data.DataLoader(ds, batch_size=128, shuffle=True)
    val = torch.utils.data.DataLoader(ds, batch_size=128, shuffle=False)
    torch.manual_seed(0)
    model = arq_uncert_fc_resnet(1, 1, 24, 3)
    net = quiet(DUN_VI, model, depth_categorical_VI([0.25] * 4, cuda=False), 300, lr=1e-2, momentum=0.9, cuda=False,
                regression=True, weight_decay=1e-4)
    out = quiet(train_fc_DUN, net, "t", str(tmp_path), 128, 6, train, val, False, seed=0, nb_its_dev=2,
                track_posterior=True, track_exact_ELBO=True, save_freq=2, q_nograd_its=2)
    elbo, nll_tr, nll_dev, err_tr, err_dev, q_app, q_true, lik, exact, basedir = out
    assert np.all(np.isfinite(elbo)) and elbo[-1] > elbo[0]          # the ELBO estimate improves
    assert err_tr[-1] < err_tr[0]
    assert q_app.shape == (6, 4) and q_true.shape == (6, 4) and exact.shape == (6,)
    assert np.allclose(q_app[0], q_app[1]) and not np.allclose(q_app[1], q_app[3])   # q frozen for 2 epochs
    assert nll_dev[0] != 0 and nll_dev[1] == 0 and nll_dev[2] != 0   # dev pass every 2nd epoch
    assert os.path.exists(os.path.join(basedir, "models", "theta_last.dat"))
    net2 = quiet(DUN_VI, arq_uncert_fc_resnet(1, 1, 24, 3), depth_categorical_VI([0.25] * 4, cuda=False), 300,
                 cuda=False, regression=True)
    quiet(net2.load, os.path.join(basedir, "models", "theta_last.dat"))
    assert net2.epoch >= 1


def test_train_fc_loop_with_exact_marginal_wrapper(cpu_backend, tmp_path):
    """DUN + depth_categorical (no q_logits) runs through train_fc_DUN as in the reference, which only touches q_logits
    when a warm-up is requested (src/DUN/train_fc.py:51-57)."""
    from dun_amd.datasets import Datafeed, load_wiggle
    from dun_amd.probability import depth_categorical
    from dun_amd.stochastic_fc_models import arq_uncert_fc_resnet
    from dun_amd.train_fc import train_fc_DUN
    from dun_amd.training_wrappers import DUN
    X, Y = load_wiggle()
    ds = Datafeed(torch.Tensor(X), torch.Tensor(Y))
    train = torch.utils.data.DataLoader(ds, batch_size=150, shuffle=False)
    torch.manual_seed(0)
    net = quiet(DUN, arq_uncert_fc_resnet(1, 1, 16, 2), depth_categorical([1 / 3.0] * 3, cuda=False), 300, lr=1e-2,
                momentum=0.9, cuda=False, regression=True)
    assert not hasattr(net.prob_model, "q_logits")
    out = quiet(train_fc_DUN, net, "t", str(tmp_path), 150, 3, train, train, False, seed=0, track_posterior=True)
    assert np.all(np.isfinite(out[0])) and out[5].shape == (3, 3) and out[6].shape == (3, 3)


# ------------------------------------------------------------------------------------------------------
# data feeding
# ------------------------------------------------------------------------------------------------------
def _write_uci_dir(z, root):
    import os
    for kind in ("standard", "gap"):
        d = os.path.join(root, "UCI_for_sharing", kind, "toy", "data")
        os.makedirs(d)
        if kind == "standard":
            np.savetxt(os.path.join(d, "data.txt"), z["table"])
            np.savetxt(os.path.join(d, "index_features.txt"), z["feature_idx"], fmt="%d")
            np.savetxt(os.path.join(d, "index_target.txt"), z["target_idx"].reshape(1), fmt="%d")
        for i in range(3):
            np.savetxt(os.path.join(d, "index_train_%d.txt" % i), z["%s_train_%d" % (kind, i)], fmt="%d")
            np.savetxt(os.path.join(d, "index_test_%d.txt" % i), z["%s_test_%d" % (kind, i)], fmt="%d")


def test_load_gap_uci_matches_reference_loader(tmp_path):
    """uci_small.npz: a synthetic table in the UCI layout and what the reference's load_gap_UCI returned for it."""
    import os
    from dun_amd.datasets import load_gap_UCI
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "uci_small.npz"))
    _write_uci_dir(z, str(tmp_path))
    names = ["X_train", "X_test", "x_means", "x_stds", "y_train", "y_test", "y_means", "y_stds"]
    for kind, gap in (("gap", True), ("standard", False)):
        for sp in (0, 2):
            res = load_gap_UCI(str(tmp_path), "toy", n_split=sp, gap=gap)
            for nm, v in zip(names, res):
                ref = z["out/%s/%d/%s" % (kind, sp, nm)]
                assert np.asarray(v).dtype == ref.dtype and np.array_equal(np.asarray(v), ref), (kind, sp, nm)
    with pytest.raises(FileNotFoundError):
        load_gap_UCI(str(tmp_path / "nowhere"), "toy")


def test_device_loader_batches_and_rank_shards():
    from dun_amd.datasets import DeviceLoader
    x = torch.arange(50, dtype=torch.float32).view(25, 2)
    y = torch.arange(25, dtype=torch.float32).view(25, 1)
    dl = DeviceLoader(x, y, batch_size=8, device="cpu")
    got = list(dl)
    assert len(dl) == 4 and len(dl.dataset) == 25 and [b[0].shape[0] for b in got] == [8, 8, 8, 1]
    assert torch.equal(torch.cat([b[1] for b in got]), y)
    # shuffled: a permutation of the rows, x and y stay aligned, epochs differ
    dl = DeviceLoader(x, y, batch_size=8, shuffle=True, device="cpu", seed=3)
    e1 = torch.cat([b[1] for b in dl]).view(-1)
    e2 = torch.cat([b[1] for b in dl]).view(-1)
    assert sorted(e1.tolist()) == list(range(25)) and not torch.equal(e1, e2)
    for xb, yb in dl:
        assert torch.equal(xb[:, 0], 2 * yb[:, 0])
    # two ranks: the halves of each global batch, same permutation on both ranks, ragged tail dropped
    parts = [list(DeviceLoader(x, y, batch_size=8, shuffle=True, device="cpu", seed=5, rank=r, world=2)) for r in (0, 1)]
    full = list(DeviceLoader(x, y, batch_size=8, shuffle=True, device="cpu", seed=5, drop_last=True))
    assert len(parts[0]) == len(parts[1]) == len(full) == 3
    for b0, b1, bf in zip(parts[0], parts[1], full):
        assert torch.equal(torch.cat([b0[1], b1[1]]), bf[1])


def test_fused_sgd_steps_channels_last_parameters(cpu_backend):
    """A conv weight in NHWC memory order is stepped in place, in its own element order, like torch.optim.SGD does."""
    from dun_amd.optim import FusedSGD
    g = torch.Generator().manual_seed(0)
    w0 = torch.randn(8, 4, 3, 3, generator=g)
    grads = [torch.randn(8, 4, 3, 3, generator=g) for _ in range(3)]
    ref = torch.nn.Parameter(w0.clone())
    mine = torch.nn.Parameter(w0.clone().contiguous(memory_format=torch.channels_last))
    o_ref = torch.optim.SGD([ref], lr=0.1, momentum=0.9, weight_decay=1e-2)
    o_mine = FusedSGD([mine], lr=0.1, momentum=0.9, weight_decay=1e-2)
    for i, gr in enumerate(grads):
        ref.grad = gr.clone()
        mine.grad = gr.clone() if i == 1 else gr.clone().contiguous(memory_format=torch.channels_last)
        o_ref.step()
        o_mine.step()
    assert mine.is_contiguous(memory_format=torch.channels_last)
    assert torch.allclose(mine.data, ref.data, rtol=1e-6, atol=1e-7)


def test_committed_bench_line_keeps_the_driver_contract():
    """The headline line bench.py printed on the GPU box (committed under profiles/) carries every field the driver and
    the judge read, with consistent values."""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = json.load(open(os.path.join(root, "profiles", "r01_bench_c3_wide.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "samples/s" and d["dtype"] == "f32" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert abs(r["achieved"] - r["flop_per_launch"] / (r["ms_per_launch"] * 1e-3) / 1e12) < 1e-6 * r["achieved"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    # value = global batch / step time
    assert abs(d["value"] - 65536 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]


def test_one_row_training_batch_raises_like_batchnorm(cpu_backend):
    """nn.BatchNorm1d refuses a training batch of one row (the reference raises this ValueError from a ragged last
    batch, src/DUN/layers.py:33-53 through torch.nn.functional.batch_norm); evaluation of one row is fine."""
    c = Case("c2_boston")
    net = build_cpu_net(c)
    x, y = c.t("x"), c.y()
    net.fit(x, y)
    rv = net.model.running_var.clone()
    with pytest.raises(ValueError, match="more than 1 value per channel"):
        net.fit(x[:1], y[:1])
    assert torch.equal(rv, net.model.running_var)  # nothing ran
    full = net.predict(x[:4])
    assert torch.allclose(net.predict(x[2:3]), full[2:3], rtol=1e-5, atol=1e-6)
    assert len(net.eval(x[:1], y[:1])) == 2


def test_file_based_toy_loaders(tmp_path):
    """load_agw_1d / load_matern_1d / load_my_1d read the .npy layouts of the reference's experiments/data directory
    (src/datasets/additional_gap_loader.py:65-223).  Parity unpinned by reference data: synthetic files of the same
    layout, the standardisation restated here; a missing file raises instead of downloading."""
    from dun_amd.datasets import load_agw_1d, load_matern_1d, load_my_1d
    rs = np.random.RandomState(7)
    base = str(tmp_path)
    for sub in ("agw_data", "matern_data", "my_1d_data"):
        os.makedirs(os.path.join(base, sub))
    agw = np.stack([rs.uniform(-8, 8, 400), rs.randn(400) * 3 + 1], axis=1)
    np.save(os.path.join(base, "agw_data", "data.npy"), agw)
    mat = rs.randn(200, 2)
    np.save(os.path.join(base, "matern_data", "matern_1d.npy"), mat)
    my = rs.randn(250, 4)
    np.save(os.path.join(base, "my_1d_data", "my_1d_data.npy"), my)

    X, Y = load_agw_1d(base)
    assert X.shape == (400, 1) and Y.shape == (400, 1) and X.dtype == Y.dtype == np.float32
    assert np.allclose(X[:, 0], (agw[:, 0] - agw[:, 0].mean()) / agw[:, 0].std(), atol=1e-6)
    assert np.allclose(Y[:, 0], (agw[:, 1] - agw[:, 1].mean()) / agw[:, 1].std(), atol=1e-6)
    assert abs(float(X.mean())) < 1e-5 and abs(float(X.std()) - 1) < 1e-5
    F, Y2 = load_agw_1d(base, get_feats=True)
    f = np.stack([agw[:, 0] / 2, (agw[:, 0] / 2) ** 2], axis=1)
    assert F.shape == (400, 2) and np.allclose(F, (f - f.mean(0)) / f.std(0), atol=1e-5) and np.array_equal(Y, Y2)
    x, y = load_matern_1d(base)
    assert x.shape == y.shape == (200, 1) and np.array_equal(x[:, 0], mat[:, 0]) and np.array_equal(y[:, 0], mat[:, 1])
    parts = load_my_1d(base)
    assert len(parts) == 4
    for i, t in enumerate(parts):
        assert t.shape == (250, 1) and t.dtype == np.float32 and np.array_equal(t[:, 0], my[:, i].astype(np.float32))
    with pytest.raises(FileNotFoundError):
        load_agw_1d(os.path.join(base, "nowhere"))


def test_wrapper_leaves_the_convolution_search_flag_alone(cpu_backend):
    """The reference turns cudnn.benchmark on in its constructor; on MIOpen that flag means the exhaustive per-shape search
    (minutes to tens of minutes on a cold box), so dun_amd's wrapper leaves it as the caller has it (training_wrappers.py
    create_net) -- and importing the image module does not touch MIOpen's environment either."""
    import subprocess
    import sys
    code = ("import os, torch; before = torch.backends.cudnn.benchmark; import dun_amd.stochastic_img_resnets; "
            "assert torch.backends.cudnn.benchmark == before; "
            "assert not [k for k in os.environ if k.startswith('MIOPEN_')], sorted(os.environ); print('ok')")
    env = {k: v for k, v in os.environ.items() if not k.startswith("MIOPEN_")}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], env=env, cwd=root, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr
    src = open(os.path.join(root, "dun_amd", "training_wrappers.py")).read()
    assert "cudnn.benchmark = True\n" not in src.replace("torch.backends.", "")
