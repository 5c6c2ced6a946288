"""Full-size parity (BASELINE configs[2] geometry: width 2048, batch 65536): one train-mode forward + fused objective +
backward of dun_amd against the oracle's functions evaluated on the same device tensors through torch's own fp32 /
fp64 kernels (the oracle is device-agnostic plain torch; on the GPU it finishes in seconds at this size), plus the
size-independent properties of the path: invariance of the per-depth sums under a permutation of the batch rows and
additivity of the likelihood sums over batch shards.
"""
import contextlib
import io

import pytest
import torch

from oracle import dun_oracle as O

pytestmark = pytest.mark.gpu

DIN, W, N_TRAIN = 64, 2048, 1 << 20


def build(L):
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU")
    import bench
    with contextlib.redirect_stdout(io.StringIO()):
        return bench.build_net(DIN, W, L, N_TRAIN, 1e-4, 0.5, 0.0)


def data(B):
    import bench
    return bench.shard_data(B, DIN, 0, torch.device("cuda"))


def ours(net, x, y):
    net.model.train()
    act = net.model.forward(x)
    prior = net.model.get_w_prior_loglike(k=None)
    out = net.prob_model.fused_objective(prior, act, y, net.f_neg_loglike, N_TRAIN)
    net._backward(act, prior, out)
    torch.cuda.synchronize()
    return act.detach(), out


def oracle_on_device(net, x, y, L, dt):
    p = {k: net.model.stacked[k].detach().to(dt).clone().requires_grad_(True) for k in O.PARAM_KEYS}
    p["running_mean"] = torch.zeros(L, W, device="cuda", dtype=dt)
    p["running_var"] = torch.ones(L, W, device="cuda", dtype=dt)
    p["num_batches_tracked"] = torch.zeros(L, dtype=torch.int64, device="cuda")
    ql = net.prob_model.q_logits.detach().to(dt).clone().requires_grad_(True)
    ls = net.f_neg_loglike.log_std.detach().to(dt).clone().requires_grad_(True)
    D = L + 1
    logprior = torch.full((D,), 1.0 / D, device="cuda", dtype=dt).log()
    a = O.fc_forward(p, x.to(dt), train=True)
    joint = O.joint_loglike(torch.zeros(D, device="cuda", dtype=dt), a, y.to(dt), True, N_TRAIN, ls)
    elbo = (O.q_probs(ql) * joint).sum() - O.kl_q_prior(ql, logprior)
    names = list(O.PARAM_KEYS) + ["q_logits", "log_std"]
    g = torch.autograd.grad(-elbo / N_TRAIN, [p[k] for k in O.PARAM_KEYS] + [ql, ls])
    return a.detach(), joint.detach(), elbo.detach(), dict(zip(names, g))


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def rel2(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def our_grads(net):
    g = {k: net.model.stacked_grad[k] for k in O.PARAM_KEYS}
    g["q_logits"] = net.prob_model.q_logits.grad
    g["log_std"] = net.f_neg_loglike.log_std.grad
    return g


def test_full_batch_four_blocks_against_fp64_and_fp32_oracle():
    L, B = 4, 65536
    net = build(L)
    x, y = data(B)
    act, out = ours(net, x, y)
    a64, j64, e64, g64 = oracle_on_device(net, x, y, L, torch.float64)
    a32, j32, e32, g32 = oracle_on_device(net, x, y, L, torch.float32)
    # north-star bar: 1e-5 relative on per-depth log-likelihood sums, ELBO (and the activations behind them)
    assert rel(act, a64) < 1e-5
    assert rel(out.joint, j64) < 1e-5 and rel(out.joint, j32) < 1e-5
    assert rel(out.objective, e64) < 1e-5 and rel(out.objective, e32) < 1e-5
    # gradients: ReLU sign coin-tosses near 0 move fp32 gradients by O(1/B); the budget is the fp32
    # oracle's own distance from the fp64 evaluation, in the l2 norm over each tensor
    g = our_grads(net)
    for k in g:
        mine, theirs = rel2(g[k], g64[k]), rel2(g32[k], g64[k])
        assert mine <= 4.0 * theirs + 2e-5, (k, mine, theirs)


def test_full_config_fifty_blocks_against_fp32_oracle():
    L, B = 50, 65536
    net = build(L)
    x, y = data(B)
    act, out = ours(net, x, y)
    a32, j32, e32, g32 = oracle_on_device(net, x, y, L, torch.float32)
    assert rel(act, a32) < 1e-5
    assert rel(out.joint, j32) < 1e-5
    assert rel(out.objective, e32) < 1e-5
    g = our_grads(net)
    assert rel(g["q_logits"], g32["q_logits"]) < 1e-5
    assert rel(g["log_std"], g32["log_std"]) < 1e-5
    assert rel(g["w_out"], g32["w_out"]) < 1e-4 and rel(g["b_out"], g32["b_out"]) < 1e-4
    for k in ("w_in", "b_in", "w", "b", "gamma", "beta"):
        # two fp32 evaluations of a 50-block ReLU/BN net: sign flips of near-zero pre-activations dominate
        assert rel2(g[k], g32[k]) < 3e-3, (k, rel2(g[k], g32[k]))


def test_batch_permutation_leaves_depth_sums_unchanged():
    L, B = 6, 65536
    net = build(L)
    x, y = data(B)
    _, out = ours(net, x, y)
    j1, e1 = out.joint.clone(), out.objective.clone()
    gw1 = net.model.stacked_grad["w"].clone()
    perm = torch.randperm(B, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))
    net2 = build(L)
    _, out2 = ours(net2, x[perm].contiguous(), y[perm].contiguous())
    assert rel(out2.joint, j1) < 2e-6
    assert rel(out2.objective, e1) < 2e-6
    assert rel2(net2.model.stacked_grad["w"], gw1) < 1e-3


def test_likelihood_sums_add_over_batch_shards():
    """Sum_b loglike[d,b] over the full batch equals the sum of the shard sums (what the N-GPU path all-reduces)."""
    from dun_amd import kernels
    be = kernels.backend()
    D, B = 51, 65536
    g = torch.Generator(device="cuda").manual_seed(11)
    act = torch.randn(D, B, 1, device="cuda", generator=g)
    y = torch.randn(B, 1, device="cuda", generator=g)
    ls = torch.tensor([-0.3], device="cuda")
    ql = torch.randn(D, device="cuda", generator=g)
    full = be.ll_pass(act, y, True, ls, ql, True, 1.0 / B, want_dact=False)[0]
    parts = None
    for r in range(8):
        sl = slice(r * B // 8, (r + 1) * B // 8)
        s = be.ll_pass(act[:, sl].contiguous(), y[sl].contiguous(), True, ls, ql, True, 1.0 / B, want_dact=False)[0]
        parts = s if parts is None else parts + s
    # fp32 only inside a thread's <= 16 elements, fp64 from there on: regrouping moves the sums by ~1e-8 relative
    assert rel(parts, full) < 1e-6


def test_full_size_image_head_and_softmax_elbo_against_fp64_on_device():
    """C5 geometry of the custom-kernel part of an image DUN: 12 exits x 4096 rows x 2048 pooled features, 10 classes.
    Batched head + fused CE/ELBO + head backward against the oracle's functions in fp64 on the same device tensors."""
    from dun_amd import kernels
    be = kernels.backend()
    D, B, K, C, N = 12, 4096, 2048, 10, 50000.0
    g = torch.Generator(device="cuda").manual_seed(5)
    feat = torch.randn(D, B, K, device="cuda", generator=g).relu_()
    w = torch.randn(C, K, device="cuda", generator=g) * 0.03
    b = torch.randn(C, device="cuda", generator=g) * 0.1
    y = torch.randint(0, C, (B,), device="cuda", generator=g)
    ql = 0.3 * torch.randn(D, device="cuda", generator=g)
    logprior = torch.full((D,), 1.0 / D, device="cuda").log()
    act = be.head_forward(feat.view(D * B, K), w, b).view(D, B, C)
    sums, dact, _ = be.ll_pass(act, y, False, None, ql, True, 1.0 / B, want_dact=True)
    res = be.depth_finalize(sums, False, ql, logprior, None, None, N, B, 1.0, 1e-35, 0)
    dw, db, dfeat = be.head_backward(feat.view(D * B, K), w, dact.view(D * B, C), need_dH=True)
    # fp64 truth
    f64, w64, b64 = feat.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    a64 = torch.nn.functional.linear(f64, w64, b64)
    z = torch.zeros(D, device="cuda", dtype=torch.float64)
    joint = O.joint_loglike(z, a64, y, False, N)
    elbo = O.elbo(z, a64, y, False, N, ql.double(), logprior.double())
    gw, gb, gf = torch.autograd.grad(-elbo / N, [w64, b64, f64])
    assert rel(act, a64.detach()) < 1e-5
    assert rel(res[8:8 + D], joint.detach()) < 1e-5 and rel(res[0], elbo.detach()) < 1e-5
    assert rel(dw, gw) < 2e-5 and rel(db, gb) < 2e-5 and rel(dfeat.view(D, B, K), gf) < 2e-5


def test_full_batch_training_trajectory_against_the_oracle_on_device():
    """Three DUN_VI.fit steps at width 2048 / batch 65536 (6 blocks): returned (ELBO estimate, NLL, RMSE) triples, the
    parameters after the steps, q and log_std, against OracleNet running the same steps through torch's fp32 kernels."""
    L, B = 6, 65536
    net = build(L)
    x, y = data(B)
    p = {k: net.model.stacked[k].detach().clone() for k in O.PARAM_KEYS}
    p0 = {k: v.clone() for k, v in p.items()}
    p["running_mean"] = torch.zeros(L, W, device="cuda")
    p["running_var"] = torch.ones(L, W, device="cuda")
    p["num_batches_tracked"] = torch.zeros(L, dtype=torch.int64, device="cuda")
    D = L + 1
    onet = O.OracleNet(p, [1.0 / D] * D, N_TRAIN, lr=1e-4, momentum=0.5, weight_decay=0.0, regression=True, vi=True)
    onet.q_logits = net.prob_model.q_logits.detach().clone()
    onet.log_std = net.f_neg_loglike.log_std.detach().clone()
    for step in range(3):
        got = net.fit(x, y)
        want = onet.fit(x, y)
        assert all(abs(g - w) <= 2e-5 * abs(w) + 1e-7 for g, w in zip(got, want)), (step, got, want)
    for k in O.PARAM_KEYS:
        # the UPDATE (three accumulated fp32 gradients) to gradient accuracy, the parameter itself to fp32 accuracy
        assert rel2(net.model.stacked[k].detach() - p0[k], onet.p[k] - p0[k]) < 3e-3, k
        assert float((net.model.stacked[k].detach() - onet.p[k]).abs().max()) < 1e-5 * float(p0[k].abs().max() + 1), k
    assert rel(net.model.running_mean, onet.p["running_mean"]) < 1e-5
    assert rel(net.model.running_var, onet.p["running_var"]) < 1e-5
    assert rel(net.prob_model.q_logits.detach(), onet.q_logits) < 1e-4
    assert rel(net.f_neg_loglike.log_std.detach(), onet.log_std) < 1e-4


def test_gemm_with_an_output_of_two_gibibytes():
    """The whole-tile GEMM kernels address their output tiles with 32-bit byte offsets from the matrix base; an output of
    2 GiB (batch 262144 at width 2048: fits a 288 GB GPU many times over) must take the flat-addressed kernel instead of
    wrapping around.  Checked on the first and the last rows, where a wrapped offset would land."""
    from dun_amd import kernels
    be = kernels.backend()
    M, N, K = 262144, 2048, 32
    g = torch.Generator(device="cuda").manual_seed(3)
    A = torch.randn(M, K, device="cuda", generator=g)
    Bw = torch.randn(N, K, device="cuda", generator=g)
    bias = torch.randn(N, device="cuda", generator=g)
    out = torch.full((M, N), float("nan"), device="cuda")
    Z, st = be.gemm(A, Bw, M, N, K, bias=bias, relu=True, colstats=True, out=out)
    assert Z.data_ptr() == out.data_ptr()
    for rows in (slice(0, 256), slice(M // 2 - 128, M // 2 + 128), slice(M - 256, M)):
        ref = torch.relu(A[rows].double() @ Bw.double().t() + bias.double())
        assert torch.isfinite(Z[rows]).all()
        assert (Z[rows].double() - ref).abs().max().item() < 1e-4
    assert bool(torch.isfinite(Z).all())
    s1 = st[0].sum(0)
    assert torch.allclose(s1, Z.double().sum(0), rtol=1e-9, atol=1e-6)
    del Z, out, st
    torch.cuda.empty_cache()
