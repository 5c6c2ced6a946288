"""GPU parity of the column-owned small-net kernels (dun_amd/csrc/smallnet.hip) behind dun_fc_forward /
dun_fc_backward, against (a) the staged C entries, which always run the tiled-GEMM kernels, and (b) an fp64
autograd evaluation of the block stack (the golden-pinned oracle covers the same path through DUN_VI.fit on the
c1/c2 fixtures in test_wrappers_gpu.py).  fp32, different summation orders: tolerances are relative to the largest magnitude of each tensor.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

# (B, din, W, L, O, residual): toy / UCI shapes, padded LDS stride (W/4 even), several row chunks, ragged batches,
# 4-float input rows (vector dW_in path), the narrowest width
SHAPES = [(300, 1, 100, 3, 1, True), (512, 13, 100, 10, 1, True), (512, 9, 128, 2, 3, True),
          (1000, 8, 64, 3, 1, False), (37, 5, 4, 2, 2, True), (2000, 13, 40, 2, 1, True), (256, 16, 100, 1, 1, True),
          # the one-launch kernels' corners: 8 column tiles, the widest batch (128 workgroups), an input layer too wide for
          # the LDS stage (din > 32), three rows (with two, BatchNorm's input gradient is pure cancellation noise), a ragged last workgroup, the narrowest width, a single column tile short of full
          (2048, 40, 128, 2, 1, True), (3, 3, 8, 2, 1, True), (17, 33, 28, 3, 1, False), (1029, 7, 112, 2, 1, True),
          (640, 64, 4, 3, 1, True), (48, 1, 124, 12, 1, True)]


def make(B, din, W, L, Oo, residual, seed=0):
    from dun_amd import kernels
    from dun_amd.stochastic_fc_models import arq_uncert_fc_MLP, arq_uncert_fc_resnet
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU")
    torch.manual_seed(seed)
    model = (arq_uncert_fc_resnet if residual else arq_uncert_fc_MLP)(din, Oo, W, L).cuda()
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        model.stacked["gamma"].copy_(1.0 + 0.2 * torch.randn(L, W, generator=g))
        model.stacked["beta"].copy_(0.2 * torch.randn(L, W, generator=g))
    x = torch.randn(B, din, generator=g).cuda()
    return model, x, kernels.backend(), g


def close(a, b, tol, what):
    a, b = a.double(), b.double()
    scale = max(float(b.abs().max()), 1e-30)
    err = float((a - b).abs().max()) / scale
    assert err <= tol, "%s: max error %.3g of the largest magnitude (tolerance %.1g)" % (what, err, tol)


def staged_forward(be, buf, L, B):
    be.fc_input_layer(buf)
    for i in range(L):
        be.fc_block_linear(buf, i)
        be.fc_block_norm(buf, i, B)
    be.fc_head(buf)


def staged_backward(be, buf, L, B):
    be.fc_bwd_begin(buf)
    for i in range(L - 1, -1, -1):
        be.fc_bwd_block_reduce(buf, i)
        be.fc_bwd_block_apply(buf, i, B)
    be.fc_bwd_end(buf)


@pytest.mark.parametrize("shape", SHAPES)
def test_small_net_path_matches_tiled_path(shape):
    from dun_amd import _lib
    B, din, W, L, Oo, residual = shape
    assert _lib.lib().dun_fc_workspace_floats(B, din, W, Oo, L) > B * W      # the shape takes the small-net path
    model, x, be, g = make(*shape)
    model.train()
    rm0, rv0 = model.running_mean.clone(), model.running_var.clone()
    b1 = be.fc_buffers(model, B, L, True)
    b1.new_act(); b1.set_input(x)
    be.fc_forward(b1)
    rm1, rv1 = model.running_mean.clone(), model.running_var.clone()
    model.running_mean.copy_(rm0); model.running_var.copy_(rv0)
    b2 = be.fc_buffers(model, B, L, True)
    b2.new_act(); b2.set_input(x)
    staged_forward(be, b2, L, B)
    close(b1.H, b2.H, 2e-4, "hidden states")   # two fp32 summation orders through BatchNorm (din = 1: near-constant columns)
    close(b1.act, b2.act, 2e-4, "act_vec")
    close(b1.Z, b2.Z, 2e-4, "ReLU(Linear)")
    close(b1.bn_mean[:L], b2.bn_mean[:L], 1e-4, "batch mean")
    close(b1.bn_rstd[:L], b2.bn_rstd[:L], 2e-4, "batch rstd")
    close(rm1, model.running_mean, 1e-4, "running mean")
    close(rv1, model.running_var, 1e-4, "running var")
    # backward from the SAME saved forward state (a ReLU pre-activation within rounding of 0 may otherwise take
    # different sides in the two forwards and move single gradient entries by O(1/B))
    for k in ("H", "Z", "bn_mean", "bn_rstd"):
        getattr(b2, k).copy_(getattr(b1, k))
    dact = (torch.randn(L + 1, B, Oo, generator=g) / B).cuda()
    b1.ensure_backward(); b1.set_dact(dact)
    model.flat_grads.fill_(float("nan"))
    be.fc_backward(b1)
    g1 = {k: v.clone() for k, v in model.stacked_grad.items()}
    b2.ensure_backward(); b2.set_dact(dact)
    model.flat_grads.fill_(float("nan"))
    staged_backward(be, b2, L, B)
    for k, v in model.stacked_grad.items():
        assert bool(torch.isfinite(g1[k]).all()), k
        close(g1[k], v, 5e-5, "gradient of " + k)
    close(b1.G, b2.G, 5e-5, "gradient w.r.t. the input layer's output")


@pytest.mark.parametrize("shape", [SHAPES[0], SHAPES[2], SHAPES[3]])
def test_small_net_forward_backward_against_fp64_oracle(shape):
    B, din, W, L, Oo, residual = shape
    model, x, be, g = make(*shape, seed=3)
    model.train()
    p = {k: v.detach().double().cpu().requires_grad_(True) for k, v in model.stacked.items()}
    buf = be.fc_buffers(model, B, L, True)
    buf.new_act(); buf.set_input(x)
    be.fc_forward(buf)
    # fp64 forward of the same block stack (src/DUN/layers.py:38-42), written out here: z = ReLU(h W^T + b),
    # BatchNorm over the batch (biased variance), residual add; shared head at every depth
    h = x.double().cpu() @ p["w_in"].t() + p["b_in"]
    acts = [h @ p["w_out"].t() + p["b_out"]]
    for i in range(L):
        z = torch.relu(h @ p["w"][i].t() + p["b"][i])
        m, v = z.mean(0), z.var(0, unbiased=False)
        n = (z - m) / torch.sqrt(v + 1e-5) * p["gamma"][i] + p["beta"][i]
        h = h + n if residual else n
        acts.append(h @ p["w_out"].t() + p["b_out"])
    act = torch.stack(acts)
    close(buf.act.cpu(), act.detach(), 1e-4, "act_vec vs fp64")
    dact = (torch.randn(L + 1, B, Oo, generator=g) / B)
    act.backward(dact.double())
    buf.ensure_backward(); buf.set_dact(dact.cuda())
    be.fc_backward(buf)
    for k, v in model.stacked_grad.items():
        close(v.cpu(), p[k].grad, 2e-3, "gradient of %s vs fp64 autograd" % k)   # ReLU-edge flips: see DESIGN.md 4


def test_small_net_eval_and_truncated_depth():
    B, din, W, L, Oo, residual = 300, 1, 100, 4, 1, True
    model, x, be, g = make(B, din, W, L, Oo, residual, seed=5)
    with torch.no_grad():
        model.running_mean.copy_(0.1 * torch.randn(L, W, generator=g))
        model.running_var.copy_(0.5 + torch.rand(L, W, generator=g))
    model.eval()
    for depth in (L, 2):
        b1 = be.fc_buffers(model, B, depth, False)
        b1.new_act(); b1.set_input(x)
        be.fc_forward(b1)
        b2 = be.fc_buffers(model, B, depth, False)
        b2.new_act(); b2.set_input(x)
        staged_forward(be, b2, depth, B)
        assert b1.act.shape[0] == depth + 1
        close(b1.act, b2.act, 2e-5, "eval act_vec at depth %d" % depth)
        close(b1.H, b2.H, 2e-5, "eval hidden states at depth %d" % depth)


@pytest.mark.parametrize("B,din,W,L,residual,depth", [(4096, 16, 512, 3, True, 3), (2048, 8, 1024, 2, False, 2),
                                                       (16384, 4, 128, 2, True, 1), (4096, 16, 512, 3, True, 2),
                                                       (16384, 8, 1024, 2, True, 2), (32768, 8, 512, 1, False, 1)])   # the last two: 128 x 256 tiles
def test_inference_epilogue_matches_staged_eval_path(B, din, W, L, residual, depth):
    """dun_fc_forward in eval mode on whole 128 x 128 tiles: BatchNorm(running stats) + residual + the head's row-dot
    inside the block GEMM's epilogue (gemm.hip gemm_epilogue_eval) against the staged entries, which write
    ReLU(Linear) and normalise it in a second pass.  Same per-element arithmetic ((z - mean) * scale + shift), so the
    hidden states agree to fp32 rounding of the GEMM; the head sums W terms in a different order (1e-5 of the
    largest magnitude, the bar of src/DUN/training_wrappers.py's predict outputs in DESIGN.md 4)."""
    model, x, be, g = make(B, din, W, L, 1, residual, seed=11)
    with torch.no_grad():
        model.running_mean.copy_(0.1 * torch.randn(L, W, generator=g))
        model.running_var.copy_(0.5 + torch.rand(L, W, generator=g))
    model.eval()
    rm0, rv0, nbt0 = model.running_mean.clone(), model.running_var.clone(), model.num_batches_tracked.clone()
    b1 = be.fc_buffers(model, B, depth, False)
    b1.new_act(); b1.set_input(x)
    b1.H.fill_(float("nan")); b1.act.fill_(float("nan"))
    be.fc_forward(b1)
    b2 = be.fc_buffers(model, B, depth, False)
    b2.new_act(); b2.set_input(x)
    staged_forward(be, b2, depth, B)
    assert b1.act.shape[0] == depth + 1 and bool(torch.isfinite(b1.act).all()) and bool(torch.isfinite(b1.H[:depth + 1]).all())
    close(b1.H[:depth + 1], b2.H[:depth + 1], 1e-5, "eval hidden states")
    close(b1.act, b2.act, 1e-5, "eval act_vec")
    # eval mode leaves the running statistics alone
    assert torch.equal(model.running_mean, rm0) and torch.equal(model.running_var, rv0)
    assert torch.equal(model.num_batches_tracked, nbt0)


def test_inference_epilogue_is_refused_on_ragged_tiles():
    """The descriptor's ev_* fields on a shape the 128 x 128 LDS-DMA kernel does not take: DUN_ERR_UNSUPPORTED, no launch."""
    import ctypes as C
    from dun_amd import _lib, kernels
    kernels.backend()
    M, N, K = 300, 100, 100
    A, Bm = torch.randn(M, K).cuda(), torch.randn(N, K).cuda()
    Cm = torch.full((M, N), 7.0).cuda()
    v = torch.ones(N).cuda()
    d = _lib.GemmDesc()
    C.memset(C.byref(d), 0, C.sizeof(d))
    d.A, d.lda, d.a_kcontig = A.data_ptr(), K, 1
    d.B, d.ldb, d.b_kcontig = Bm.data_ptr(), K, 1
    d.C, d.ldc = Cm.data_ptr(), N
    d.M, d.N, d.K = M, N, K
    d.bias, d.relu = v.data_ptr(), 1
    d.ev_mean = d.ev_scale = d.ev_shift = v.data_ptr()
    rc = _lib.lib().dun_gemm_f32(C.byref(d), None)
    assert rc == 3, rc   # DUN_ERR_UNSUPPORTED (include/dun_hip.h)
    torch.cuda.synchronize()
    assert bool((Cm == 7.0).all())


@pytest.mark.parametrize("shape", [SHAPES[0], SHAPES[1], SHAPES[3]])
def test_the_three_small_net_paths_agree(shape):
    """dun_fc_small_net_mode: 2 = one-launch forward / backward (persist.hip, the default), 1 = one column-owned launch
    per block (smallnet.hip), 0 = the tiled GEMM path (what DUN_SMALLNET=0 selects).  Same entries, same buffers, three
    kernel families: forward state and every gradient agree, and a whole fit step through the wrapper gives the same
    returned numbers in each mode."""
    from dun_amd import _lib
    B, din, W, L, Oo, residual = shape
    lib = _lib.lib()
    prev = lib.dun_fc_small_net_mode(-1)
    import os
    want = 0 if os.environ.get("DUN_SMALLNET", "")[:1] == "0" else (1 if os.environ.get("DUN_SMALLNET_PERSIST", "")[:1] == "0" else 2)
    assert prev == want                                           # the default is the one-launch family; the environment's choice otherwise
    outs = {}
    try:
        for mode in (2, 1, 0):
            assert lib.dun_fc_small_net_mode(mode) in (0, 1, 2) and lib.dun_fc_small_net_mode(-1) == mode
            model, x, be, g = make(*shape, seed=9)
            model.train()
            buf = be.fc_buffers(model, B, L, True)
            buf.new_act(); buf.set_input(x)
            be.fc_forward(buf)
            dact = (torch.randn(L + 1, B, Oo, generator=g) / B).cuda()
            buf.ensure_backward(); buf.set_dact(dact)
            model.flat_grads.fill_(float("nan"))
            be.fc_backward(buf)
            torch.cuda.synchronize()
            outs[mode] = dict(H=buf.H.clone(), act=buf.act.clone(), rm=model.running_mean.clone(), rv=model.running_var.clone(),
                              nbt=model.num_batches_tracked.clone(), grads={k: v.clone() for k, v in model.stacked_grad.items()})
        for mode in (1, 0):
            a, b = outs[mode], outs[2]
            close(a["H"], b["H"], 2e-4, "hidden states, mode %d" % mode)
            close(a["act"], b["act"], 2e-4, "act_vec, mode %d" % mode)
            close(a["rm"], b["rm"], 1e-4, "running mean, mode %d" % mode)
            close(a["rv"], b["rv"], 1e-4, "running var, mode %d" % mode)
            assert torch.equal(a["nbt"], b["nbt"]) and int(a["nbt"][0]) == 1
            for k, v in b["grads"].items():
                assert bool(torch.isfinite(a["grads"][k]).all()), (mode, k)
                # (each mode's backward starts from its own forward: a ReLU input within rounding of 0 may fall on
                #  either side, which moves single entries by O(1/B) -- see test_small_net_path_matches_tiled_path)
                close(a["grads"][k], v, 2e-3, "gradient of %s, mode %d" % (k, mode))
    finally:
        lib.dun_fc_small_net_mode(prev)
    assert lib.dun_fc_small_net_mode(-1) == prev


def test_fit_step_is_the_same_in_every_small_net_mode():
    from dun_amd import _lib
    from tests.util import Case
    from tests.test_wrappers_gpu import build_net
    lib = _lib.lib()
    prev = lib.dun_fc_small_net_mode(-1)
    c = Case("c2_boston")
    x, y = c.t("x"), c.y()
    rets = {}
    try:
        for mode in (2, 1, 0):
            lib.dun_fc_small_net_mode(mode)
            net = build_net(c)
            rets[mode] = np.array([net.fit(x, y) for _ in range(2)], dtype=np.float64)
            assert np.allclose(rets[mode][0], c["fit/ret"][0], rtol=2e-5, atol=1e-6), (mode, rets[mode][0])
    finally:
        lib.dun_fc_small_net_mode(prev)
    assert np.allclose(rets[1], rets[2], rtol=1e-4) and np.allclose(rets[0], rets[2], rtol=1e-4)
