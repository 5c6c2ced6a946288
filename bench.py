#!/usr/bin/env python
"""Headline benchmark: DUN ELBO training throughput (samples/s) on the wide residual-MLP DUN of BASELINE.json
configs[2] -- 50 blocks, width 2048, global batch 65536, regression, DUN_VI.fit (forward_get_acts -> fused
head/ELBO -> backward -> fused SGD -> fused metrics) -- on N MI355X, plus the roofline position of the dominant
kernel (fp32 MFMA trunk GEMM) and of the HBM-streaming head / ELBO kernels, next to the CPU oracle timed on the
box's host cores.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

The global batch is fixed (strong scaling): rank r trains on rows [r*B/N, (r+1)*B/N) with synchronised BatchNorm
statistics, so every N computes the same step as one device on the full batch.  Prints ONE JSON line on rank 0.
"""
import argparse
import contextlib
import io
import json
import math
import os
import sys
import time

# image workloads: MIOpen's default (hybrid) find measures the applicable solvers once per convolution shape -- about
# four minutes of warm-up for the 12-exit ResNet-50 at batch 4096 on a cold box -- so the benchmark asks for the FAST mode,
# which picks from heuristics without measuring (set before MIOpen loads; an explicit setting wins).  Those heuristics
# pick badly on gfx950 for fp32: the Winograd assembly solver (vector ALUs, 11.9 ms per call, half of the step) for the
# 3x3 convolutions at batch 4096, and the rocBLAS-GEMM solver for the 1x1 ones at batch 256, where the measured find
# lands on the implicit-GEMM MFMA kernels; with both solver families out of the selection FAST picks what the find
# picks: c5_cifar 2845 -> 1578 ms per step (hybrid find: 1580), c4_mnist 113.1 -> 102.4 (hybrid find, which keeps
# Winograd and GEMM where they measure faster: 97.4).  `MIOPEN_FIND_MODE=DYNAMIC_HYBRID python bench.py ...` gives those.
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")
if os.environ["MIOPEN_FIND_MODE"] == "FAST":
    os.environ.setdefault("MIOPEN_DEBUG_CONV_WINOGRAD", "0")
    os.environ.setdefault("MIOPEN_DEBUG_CONV_GEMM", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# torch is imported by the ranks only (import_torch(), called from run_rank): the parent of a self-launched N-rank job
# starts the ranks as child processes and must never initialise the GPU itself
def import_torch():
    global torch
    import torch as _torch
    torch = _torch
    return _torch


class _TorchOnFirstUse(object):
    """Stands in for the module until something needs it (helpers of this file imported by tests / scripts)."""

    def __getattr__(self, name):
        return getattr(import_torch(), name)


torch = _TorchOnFirstUse()

# MI355X peaks (MI355X_MICROARCH.md): HBM3E 8 TB/s; dense fp32 matrix (v_mfma_f32_32x32x2_f32) 157.3 TFLOP/s
HBM_PEAK_GBS = 8000.0
MFMA_F32_PEAK_TF = 157.3

WORKLOADS = {
    # name: (din, width, blocks, global batch, N_train, lr, momentum, wd)
    # lr: BASELINE.json does not fix it; the wrapper default 1e-2 (and 1e-3) diverges within three steps at 50
    # blocks in the reference arithmetic itself (checked with the CPU oracle), 1e-4 trains
    "c3_wide": (64, 2048, 50, 65536, 1 << 20, 1e-4, 0.5, 0.0),
    "c2_uci": (13, 100, 10, 512, 455, 1e-2, 0.5, 0.0),
    "c3_small": (64, 2048, 8, 8192, 1 << 20, 1e-4, 0.5, 0.0),
}


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def shard_data(B_local, din, rank, device):
    """Synthetic regression rows of this rank: x ~ N(0,1), y = tanh(x w) + 0.1 eps."""
    g = torch.Generator().manual_seed(1234)
    w = torch.randn(din, 1, generator=g) / math.sqrt(din)
    g = torch.Generator().manual_seed(rank)
    x = torch.randn(B_local, din, generator=g)
    y = torch.tanh(x @ w) + 0.1 * torch.randn(B_local, 1, generator=g)
    return x.to(device), y.to(device)


def build_net(din, W, L, N_train, lr, momentum, wd, cuda=True):
    from dun_amd.probability import depth_categorical_VI
    from dun_amd.stochastic_fc_models import arq_uncert_fc_resnet
    from dun_amd.training_wrappers import DUN_VI
    torch.manual_seed(0)
    model = arq_uncert_fc_resnet(din, 1, W, L)
    pm = quiet(depth_categorical_VI, [1.0 / (L + 1)] * (L + 1), cuda=cuda)
    net = quiet(DUN_VI, model, pm, N_train, lr=lr, momentum=momentum, weight_decay=wd, cuda=cuda, regression=True)
    g = torch.Generator().manual_seed(7)
    pm.q_logits.data.copy_(0.1 * torch.randn(L + 1, generator=g))
    return net


IMAGE_WORKLOADS = {
    # name: (channels, side, global batch, N_train)   ResNet-50 DUN, start_depth=1, end_depth=13 (D=12), 10 classes,
    # SGD(lr 0.1, momentum 0.9, wd 1e-4): experiments/image/train_DUN.py:32,45-54 in the reference
    "c4_mnist": (1, 28, 256, 60000),
    "c5_cifar": (3, 32, 4096, 50000),
}


def build_image_net(channels, N_train):
    from dun_amd.probability import depth_categorical_VI
    from dun_amd.stochastic_img_resnets import resnet50
    from dun_amd.training_wrappers import DUN_VI
    torch.manual_seed(0)
    model = resnet50(arch_uncert=True, start_depth=1, end_depth=13, num_classes=10, zero_init_residual=True,
                     initial_conv="1x3", concat_pool=False, input_chanels=channels, p_drop=0)
    D = model.n_layers
    pm = quiet(depth_categorical_VI, [1.0 / D] * D, cuda=True)
    net = quiet(DUN_VI, model, pm, N_train, lr=0.1, momentum=0.9, weight_decay=1e-4, cuda=True, regression=False)
    # never MIOpen's exhaustive per-shape search unless --miopen-benchmark asks for it (more than 18 minutes for the first
    # step of this net at batch 4096 on a cold box); the find mode is the environment's (FAST by default here, see the top
    # of this file).  The conv trunk is not ours.
    torch.backends.cudnn.benchmark = False
    return net


def image_head_rooflines(net, B_local, reps=20):
    """The custom-kernel part of an image DUN step: batched head over [D*B, 2048] features + fused CE/ELBO."""
    from dun_amd import kernels
    be = kernels.backend()
    D, K, C = net.model.n_layers, net.model.output_block.in_features, net.model.num_classes
    dev = net.model.output_block.weight.device
    feat = torch.randn(D * B_local, K, device=dev).relu()
    w, b = net.model.output_block.weight.detach().contiguous(), net.model.output_block.bias.detach().contiguous()
    out = {}
    ms = timed(lambda: be.head_forward(feat, w, b), reps)
    out["head_fwd"] = dict(ms=ms, gbs=4.0 * D * B_local * K / ms / 1e6)
    dact = torch.randn(D * B_local, C, device=dev)
    ms = timed(lambda: be.head_backward(feat, w, dact, need_dH=True), reps)
    out["head_bwd_with_dfeat"] = dict(ms=ms, gbs=8.0 * D * B_local * K / ms / 1e6)
    ql = net.prob_model.q_logits.detach()
    from dun_amd import _lib as L
    lib = L.lib()

    def elbo_point(Bn, n):
        """fused CE + per-depth sums + dact in one sweep, bare C-ABI call on preallocated buffers: 8*D*B*C + 8*B bytes"""
        act = torch.randn(D, Bn, C, device=dev)
        y = torch.randint(0, C, (Bn,), device=dev)
        dact_o = torch.empty_like(act)
        nparts = lib.dun_ll_num_partials(Bn, C)
        partials = torch.empty(D, nparts, device=dev, dtype=torch.float64)
        sp = L.stream_ptr(dev)
        ms_ = timed(lambda: L.check(lib.dun_ll_softmax(L.f32(act), L.ptr(y), L.f32(ql), 1, D, Bn, C, 1.0 / Bn,
                                                       L.f32(dact_o), None, L.ptr(partials), sp)), n)
        return dict(ms=ms_, gbs=(8.0 * D * Bn * C + 8.0 * Bn) / ms_ / 1e6, B=Bn)

    out["elbo_softmax"] = elbo_point(B_local, 50)      # the workload's own size: cache resident, latency bound
    out["elbo_softmax_2e20"] = elbo_point(1 << 20, 10)  # HBM-resident size of the same kernel
    return out


def timed(fn, n, stream_sync=True, settle_ms=60.0):
    """Average duration (ms) of n back-to-back calls, HIP events on the current stream, taken once the launch has
    been repeating for about settle_ms: after an idle gap (a host-side allocation, a first use of the RNG) the first
    launches run at lower clocks -- the C3 dX GEMM measured 4.9, 4.7, 4.4, 4.3, 4.2, 4.1 ms before settling at 3.94
    (profiles/r02_dx_first_launches.txt) -- and a five-launch average taken there reports the ramp, not the kernel."""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    torch.cuda.synchronize()
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    for _ in range(max(1, min(200, int(settle_ms / max(e0.elapsed_time(e1), 1e-3))))):
        fn()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def gemm_kernel_name(M, N, K, a_kc, b_kc, plain_few_tiles=False):
    """Which GEMM kernel the library dispatches for this shape (dun_amd/csrc/gemm.hip: gemm_f32)."""
    small = ((M + 127) // 128) * ((N + 127) // 128) < 128
    if not small and M % 128 == 0 and N % 128 == 0 and K % 32 == 0:
        from dun_amd import _lib
        if _lib.lib().dun_gemm_precision(-1) == 1:
            return "gemm_bf16x3_kernel<A_KC=%d,B_KC=%d> (EXPLORATORY: 6 bf16 MFMA products per fp32 product)" % (a_kc, b_kc)
        wide = N % 256 == 0 and (M // 128) * (N // 256) >= 512 and not plain_few_tiles and (a_kc or not b_kc)
        if wide:
            return "gemm_f32_wide_kernel<A_KC=%d,B_KC=%d> (128x256 tiles)" % (a_kc, b_kc)
        if plain_few_tiles and N % 256 == 0 and (a_kc or not b_kc) and (M // 128) * (N // 256) <= 256:
            return "gemm_f32_wide_kernel<A_KC=%d,B_KC=%d> (128x256 tiles, K ranges filling 512 workgroups)" % (a_kc, b_kc)
        return "gemm_f32_dma4_kernel<A_KC=%d,B_KC=%d>" % (a_kc, b_kc)
    mi = 1 if small else 2
    return "gemm_f32_kernel<A_KC=%d,B_KC=%d,MI=%d> (%dx%d tiles)" % (a_kc, b_kc, mi, 64 * mi, 64 * mi)


def small_net_path(B, W):
    """Mirror of sn_shape_ok (dun_amd/csrc/smallnet.hip): does dun_fc_forward/backward leave the tiled-GEMM path for
    this shape (mode 1: one launch per block, mode 2: one launch per pass; dun_fc_small_net_mode)?"""
    from dun_amd import _lib
    return _lib.lib().dun_fc_small_net_mode(-1) != 0 and 4 <= W <= 128 and W % 4 == 0 and 1 <= B <= 2048


def small_net_rooflines(net, B_local, reps):
    """Small nets: the step's trunk is ONE launch forward and ONE backward + one weight-gradient launch (persist.hip;
    latency bound: a chain of per-block BatchNorm exchanges).  Timed live on the step's own buffers through the C
    entries; flop rates are reported for scale only."""
    from dun_amd import _lib, kernels
    be = kernels.backend()
    model = net.model
    W, L = model.width, model.n_layers
    buf = model._get_buffers(B_local, L, True)
    buf.ensure_backward()
    if buf.dact is None:
        buf.set_dact(torch.zeros(L + 1, B_local, 1, device=buf.H.device))
    rm, rv, nbt = model.running_mean.clone(), model.running_var.clone(), model.num_batches_tracked.clone()
    fwd = timed(lambda: be.fc_forward(buf), reps)
    bwd = timed(lambda: be.fc_backward(buf), reps)
    model.running_mean.copy_(rm)
    model.running_var.copy_(rv)
    model.num_batches_tracked.copy_(nbt)
    mode = _lib.lib().dun_fc_small_net_mode(-1)
    f_fwd = 2.0 * B_local * (L * W * W + model.input_dim * W + (L + 1) * W)
    how = ("pn_forward_kernel: whole dun_fc_forward in ONE launch (row-owning workgroups, f32 MFMA 16x16x4 tiles, tagged-"
           "record BatchNorm exchange)" if mode == 2 and model.output_dim == 1 else
           "sn_block_fwd_kernel x%d (+ input GEMM, head): whole dun_fc_forward, column-owned block kernels" % L)
    return {
        "gemm_fwd": dict(kernel=how, ms=fwd, tflops=f_fwd / fwd / 1e9),
        "trunk_backward": dict(kernel="whole dun_fc_backward (%s)" % ("pn_backward_kernel + pn_dw_kernel" if mode == 2 else
                                                                      "sn_block_bwd_kernel x%d" % (L + 1)),
                               ms=bwd, tflops=2.0 * f_fwd / bwd / 1e9),
    }


def gemm_in_step(net, x, y, B_local):
    """ONE real training step with every GEMM launch bracketed by HIP events on its own stream (dun_gemm_profile_begin /
    _end): per-launch milliseconds of the trunk's three GEMM classes as the step runs them.  -> dict or None."""
    import ctypes as C
    from dun_amd import _lib
    lib = _lib.lib()
    W, L = net.model.width, net.model.n_layers
    cap = 4 * (3 * L + 8)
    if lib.dun_gemm_profile_begin(cap) != 0:
        return None
    try:
        net.fit(x, y)
    finally:
        recs = (_lib.GemmProfileRec * cap)()
        n = lib.dun_gemm_profile_end(recs, cap)
    cls = {"fwd": [], "dx": [], "dw": []}
    for r in recs[:n]:
        if r.M == B_local and r.N == W and r.K == W and (r.epilogue & 2):
            cls["fwd"].append(r.ms)
        elif r.M == B_local and r.N == W and r.K == W and (r.epilogue & (4 | 8)):
            cls["dx"].append(r.ms)
        elif r.M == W and r.N == W and r.K == B_local:
            cls["dw"].append(r.ms)
    flops = 2.0 * B_local * W * W
    out = {}
    for k, v in cls.items():
        if v:
            ms = sum(v) / len(v)
            out[k] = {"launches": len(v), "ms_per_launch": ms, "tflops": flops / ms / 1e9,
                      "frac": flops / ms / 1e9 / MFMA_F32_PEAK_TF}
    return out or None


def secondary_c2(steps=500, warmup=30):
    """BASELINE.json configs[1] (UCI-shaped net, batch 512) timed by the same process: value and ms/step."""
    din, W, L, B, N_train, lr, momentum, wd = WORKLOADS["c2_uci"]
    net = build_net(din, W, L, N_train, lr, momentum, wd)
    x, y = shard_data(B, din, 0, torch.device("cuda", torch.cuda.current_device()))
    for _ in range(warmup):
        last = net.fit(x, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        last = net.fit(x, y)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"workload": "c2_uci: regression DUN_VI.fit, residual MLP %d blocks x width %d, din %d, batch %d, N_train %d, "
                        "SGD(lr %g, momentum %g); whole step captured as one hipGraph" % (L, W, din, B, N_train, lr, momentum),
            "metric": "DUN ELBO train samples/sec", "value": B * steps / dt, "unit": "samples/s",
            "ms_per_step": dt / steps * 1e3, "steps": steps, "warmup": warmup, "finite": all(math.isfinite(v) for v in last)}


def kernel_rooflines(net, B_local, reps, y=None):
    """Live per-kernel timings on the buffers of the training step (same shapes, same data)."""
    from dun_amd import kernels
    be = kernels.backend()
    model = net.model
    W, L, D = model.width, model.n_layers, model.n_layers + 1
    if small_net_path(B_local, W):
        return small_net_rooflines(net, B_local, reps)
    buf = model._get_buffers(B_local, L, True)
    buf.ensure_backward()
    out = {}
    H0, Z0, w0, b0 = buf.H[0], buf.Z[0], model.stacked["w"][0], model.stacked["b"][0]
    flops = 2.0 * B_local * W * W
    scratch = torch.empty_like(Z0)
    ms = timed(lambda: be.gemm(H0, w0, B_local, W, W, bias=b0, relu=True, colstats=True, out=scratch), reps)
    out["gemm_fwd"] = dict(kernel=gemm_kernel_name(B_local, W, W, 1, 1) + " (block forward: bias+ReLU+BN stats)",
                           ms=ms, tflops=flops / ms / 1e9)
    u1 = torch.randn(B_local, 1, device=Z0.device) * 1e-3
    v1 = model.stacked["w_out"]
    # dX on the operands the step left behind: A = dU of the last block processed, running gradient updated in place
    # and in the form the step launches: against the transposed copy of the block weights (both operands K-contiguous),
    # residual seeding the accumulators, BatchNorm backward column statistics of the block below in the epilogue
    # (out of place here -- same tile reads and writes as the step's in-place update, but repeated launches do not keep
    # accumulating into their own input)
    w0t = w0.t().contiguous()
    bn = (Z0, buf.bn_mean[0], buf.bn_rstd[0])
    ms = timed(lambda: be.gemm(buf.dU, w0t, B_local, W, W, a_kcontig=True, b_kcontig=True, R=buf.G, U=u1, V=v1,
                               out=scratch, bn=bn), reps)
    out["gemm_dx"] = dict(kernel=gemm_kernel_name(B_local, W, W, 1, 1) + " (dX = dU W + g + dact w_out in place, + BN-backward "
                                 "column sums of the block below)", ms=ms, tflops=flops / ms / 1e9)
    gw = model.stacked_grad["w"][0]
    skws = torch.empty(2 * W * W, device=Z0.device)
    ms = timed(lambda: be.gemm(buf.dU, H0, W, W, B_local, a_kcontig=False, b_kcontig=False, out=gw, splitk_ws=skws),
               reps)
    out["gemm_dw"] = dict(kernel=gemm_kernel_name(W, W, B_local, 0, 0) + ", split-K + splitk_sum (dW = dU^T H)", ms=ms, tflops=flops / ms / 1e9)
    # head over all D*B hidden rows (O=1): reads 4*D*B*W bytes once
    Hall = buf.H.view(D * B_local, W)
    ms = timed(lambda: be.head_forward(Hall, model.stacked["w_out"], model.stacked["b_out"]), reps)
    out["head_fwd"] = dict(ms=ms, gbs=4.0 * D * B_local * W / ms / 1e6)
    dact = torch.randn(D * B_local, 1, device=Hall.device)
    ms = timed(lambda: be.head_backward(Hall, model.stacked["w_out"], dact, need_dH=False), reps)
    out["head_bwd"] = dict(ms=ms, gbs=4.0 * D * B_local * W / ms / 1e6)
    # What the step really runs for the head and the objective: BatchNorm apply + residual + head row-dot + Gaussian
    # likelihood terms (act, dact, per-depth partial sums) in ONE row-owned pass over a [B,W] plane (dun_fc_block_norm:
    # a W-element statistics kernel + bn_apply_head_kernel).  Algorithmic bytes: Z and H_in read, H_out written
    # (12*B*W) + y read, act and dact written (12*B).  Timed: the streaming kernel alone (dun_fc_block_apply).
    if y is not None and buf.head_fused:
        rm, rv = model.running_mean.clone(), model.running_var.clone()
        spec = net.prob_model.head_objective(y, net.f_neg_loglike, B_local)
        buf.new_act()
        buf.set_objective(spec)
        be.fc_block_norm(buf, 0, float(B_local))  # (statistics of block 0 into bn_mean / bn_rstd)
        ms = timed(lambda: be.fc_block_apply(buf, 0), reps)
        buf.set_objective(None)
        model.running_mean.copy_(rm)
        model.running_var.copy_(rv)
        out["head_elbo_fused"] = dict(ms=ms, gbs=(12.0 * B_local * W + 12.0 * B_local) / ms / 1e6)
    # fused ELBO pass (Gaussian likelihood + per-depth reduction + dact in one sweep) on act_vec [D,B]:
    # 8*D*B + 4*B algorithmic bytes.  Timed as the bare C-ABI call on preallocated buffers.
    from dun_amd import _lib as L
    lib = L.lib()
    ls = net.f_neg_loglike.log_std.detach()
    ql = net.prob_model.q_logits.detach()

    def elbo_point(Bn, n):
        act = torch.randn(D, Bn, device=Hall.device)
        yv = torch.randn(Bn, device=Hall.device)
        dact_o = torch.empty_like(act)
        nparts = lib.dun_ll_num_partials(Bn, 1)
        partials = torch.empty(D, nparts, device=Hall.device, dtype=torch.float64)
        sp = L.stream_ptr(Hall.device)
        ms_ = timed(lambda: L.check(lib.dun_ll_gauss(L.f32(act), L.f32(yv), L.f32(ls), L.f32(ql), 1, D, Bn, 1.0 / Bn,
                                                     L.f32(dact_o), None, L.ptr(partials), sp)), n)
        return dict(ms=ms_, gbs=(8.0 * D * Bn + 4.0 * Bn) / ms_ / 1e6, B=Bn)

    out["elbo"] = elbo_point(B_local, 50)          # the workload's own act_vec: cache resident, launch bound
    # the same kernel at sizes that leave the 256 MiB Infinity Cache (report a size sweep)
    out["elbo_2e20"] = elbo_point(1 << 20, 20)
    out["elbo_2e22"] = elbo_point(1 << 22, 10)
    return out


def pmc_traffic(kernel, workload, world):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json:
    2*FETCH_SIZE + WRITE_SIZE in bytes, gfx950 correction applied), valid for the single-GPU shapes it was taken on."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if world != 1 or not os.path.exists(path):
        return None
    try:
        return json.load(open(path)).get(workload, {}).get(kernel)
    except Exception:
        return None


def cpu_baseline(din, W, L, N_train, lr, momentum, wd, B_cpu):
    """The oracle (CPU restatement of the reference path, pinned to the reference by tests/golden) on host cores."""
    from oracle import dun_oracle as O
    threads = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "0") or 0) or (os.cpu_count() or 1), 64)
    torch.set_num_threads(threads)
    g = torch.Generator().manual_seed(0)
    p = O.init_fc_params(din, 1, W, L, generator=g)
    net = O.OracleNet(p, [1.0 / (L + 1)] * (L + 1), N_train, lr=lr, momentum=momentum, weight_decay=wd,
                      regression=True, vi=True)
    x, y = O.synthetic_regression(B_cpu, din, seed=0)
    net.fit(x, y)  # warm-up (thread pools, allocator)
    t0 = time.time()
    steps = 0
    # a bounded sample: about 10 s of CPU work (two steps of the wide net, a few thousand of the UCI one), at least 1 step
    while steps < 1 or (time.time() - t0 < 10.0 and steps < 20000):
        net.fit(x, y)
        steps += 1
    dt = (time.time() - t0) / steps
    return dict(value=B_cpu / dt, unit="samples/s", cores=threads, kind="port",
                sample="%d DUN_VI.fit step(s) of the CPU oracle at batch %d (full workload batch: see config), "
                       "%.4g s/step, %.1f s of CPU work, throughput scaled per sample" % (steps, B_cpu, dt, dt * steps))


def label_exploratory(result):
    """EXPLORATORY switch (DUN_GEMM_BF16X3=1 / dun_gemm_precision(1)): never the headline, labelled as what it is."""
    from dun_amd import _lib
    if _lib.lib().dun_gemm_precision(-1) != 1:
        return
    result["dtype"] = ("f32 operands split into 3 bf16 digits, 6 bf16 MFMA products, f32 accumulate "
                       "(EXPLORATORY switch, not the measured path)")
    result["metric"] = "EXPLORATORY (bf16x3 GEMMs): " + result["metric"]
    if "roofline" in result:
        result["roofline"]["note"] = ("exploratory bf16x3 kernel: `achieved` counts the fp32-equivalent 2*M*N*K flop, "
                                      "`peak` stays the exact-fp32 MFMA peak for scale -- not a roofline claim")


def _claim_stdout():
    """The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints a version banner to fd 1 when
    its first communicator comes up), so fd 1 is pointed at stderr for the whole run and the line goes to the original
    stdout, returned here as a file object."""
    sys.stdout.flush()
    out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    return out


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None,
                    help="timed steps (default 5; 500 for the small-net workload c2_uci, whose step is ~0.13 ms)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps before them (default 2; 30 for c2_uci)")
    ap.add_argument("--workload", default="c3_wide", choices=sorted(WORKLOADS) + sorted(IMAGE_WORKLOADS))
    ap.add_argument("--no-sync-bn", action="store_true", help="per-replica BatchNorm statistics (not matched numerics)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-rooflines", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the c2_uci figure the c3_wide line carries")
    ap.add_argument("--cpu-batch", type=int, default=0)
    ap.add_argument("--rehearse-dist", action="store_true",
                    help="one-GPU rehearsal of the N>1 branch of this script: a one-rank RCCL group, the sharded step "
                         "with every collective issued, the barriers and the MAX-over-ranks timing")
    ap.add_argument("--cpu-rehearsal", action="store_true",
                    help="TEST SEAM, not a measurement: run the launch / rendezvous / sharded-step / JSON plumbing of "
                         "this script on CPU ranks (gloo) with the tests' stand-in backend (tests/cpu_backend.py) on a "
                         "tiny net; the line it prints is labelled as a rehearsal")
    ap.add_argument("--launch-timeout", type=float, default=1500.0,
                    help="self-launched N-rank job: seconds after which the parent ends ranks that are still running")
    ap.add_argument("--predict", action="store_true",
                    help="time DUN_VI.predict(x, get_std=True) (eval-mode forward + depth-marginal predictor) instead "
                         "of fit; prints the same JSON line with metric 'DUN predict samples/sec'")
    ap.add_argument("--miopen-benchmark", action="store_true",
                    help="image workloads: let MIOpen search per conv shape (torch.backends.cudnn.benchmark, as the "
                         "reference's wrapper sets it) instead of immediate mode; on a cold box the search of the 12-exit "
                         "ResNet-50 at batch 4096 had not finished its first step after 18 minutes (round 3) -- a line a "
                         "minute on stderr says the run is alive")
    ap.add_argument("--channels-last", action="store_true",
                    help="image workloads: NHWC conv trunk; the head/ELBO path is unaffected (measured SLOWER with MIOpen's "
                         "FAST find mode on fp32: 13.0 s vs 1.58 s per c5_cifar step; with MIOPEN_FIND_MODE=DYNAMIC_HYBRID "
                         "1.80 s vs 1.58 s, MIOpen's NHWC BatchNorm costing more than the transposes saved -- off by default)")
    args = ap.parse_args(argv)
    small = args.workload == "c2_uci" and not args.predict
    if args.steps is None:
        args.steps = 500 if small else 5
    if args.warmup is None:
        args.warmup = 30 if small else 2
    return args


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes of this same script (one per GPU,
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment, exactly what torch.distributed.run would set),
    relay rank 0's JSON line, and fail if any rank fails.  This parent process never imports torch or touches a GPU."""
    import subprocess
    n = args.gpus
    port = _free_port()
    procs = []
    import tempfile
    out0 = tempfile.TemporaryFile()  # rank 0's stdout: a file, not a pipe (a pipe nobody drains blocks the writer at 64 KiB)
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), DUN_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # one JSON line comes from rank 0's stdout; what the other ranks might print there goes to our stderr
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=out0 if r == 0 else sys.stderr, stderr=sys.stderr))
    deadline = time.time() + args.launch_timeout
    failed = None
    live = set(range(n))
    while live and failed is None:
        for r in sorted(live):
            rc = procs[r].poll()
            if rc is None:
                continue
            live.discard(r)
            if rc != 0:
                failed = (r, rc)
                break
        if live and failed is None:
            if time.time() > deadline:
                failed = (min(live), "timeout after %.0f s" % args.launch_timeout)
                break
            time.sleep(0.2)
    if failed is not None:
        for r in sorted(live):  # the exact processes started above, nothing else
            procs[r].terminate()
        t_end = time.time() + 20
        for r in sorted(live):
            try:
                procs[r].wait(timeout=max(0.1, t_end - time.time()))
            except subprocess.TimeoutExpired:
                procs[r].kill()
        print("bench.py: rank %d of %d failed (%s); no result" % (failed[0], n, failed[1]), file=sys.stderr, flush=True)
        return 1
    out0.seek(0)
    out = out0.read().decode()
    lines = [ln for ln in out.splitlines() if ln.strip().startswith("{")]
    if len(lines) != 1:
        print("bench.py: expected one JSON line from rank 0, got %d" % len(lines), file=sys.stderr, flush=True)
        return 1
    print(lines[0], flush=True)
    return 0


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.rehearse_dist:
        sys.exit(launch_ranks(args, argv))
    run_rank(args)


def run_rank(args):
    json_out = _claim_stdout()
    import_torch()
    cpu = bool(args.cpu_rehearsal)
    if cpu:
        from dun_amd import kernels
        from tests.cpu_backend import CpuBackend
        kernels.set_backend(CpuBackend())
        WORKLOADS["cpu_rehearsal"] = (8, 32, 3, 64, 1 << 10, 1e-3, 0.5, 0.0)
        args.workload = "cpu_rehearsal"
        args.no_kernel_rooflines = args.no_cpu_baseline = True

    image = args.workload in IMAGE_WORKLOADS
    if image:
        channels, side, B, N_train = IMAGE_WORKLOADS[args.workload]
        lr, momentum, wd = 0.1, 0.9, 1e-4
    else:
        din, W, L, B, N_train, lr, momentum, wd = WORKLOADS[args.workload]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and not args.rehearse_dist:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d in the environment" % (args.gpus, world))
    if cpu:
        device = torch.device("cpu")
        torch.set_num_threads(2)
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the DUN kernels have no CPU path")
        if local_rank >= torch.cuda.device_count():
            raise SystemExit("bench.py: rank %d has no GPU (LOCAL_RANK %d, %d visible)" % (rank, local_rank, torch.cuda.device_count()))
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
    dist_on = world > 1 or args.rehearse_dist
    world_formed = ranks_counted = 1
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse_dist:
            for k, v in (("MASTER_PORT", "29541"), ("RANK", "0"), ("WORLD_SIZE", "1")):
                os.environ.setdefault(k, v)
        if cpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)
        # did the communicator really form over N ranks?  its own size, and a SUM all-reduce of ones over the wire
        world_formed = dist.get_world_size()
        ones = torch.ones(1, device=device, dtype=torch.float64)
        dist.all_reduce(ones)
        ranks_counted = int(round(float(ones.item())))
        if world_formed != world or ranks_counted != world:
            raise SystemExit("bench.py: asked for %d ranks, the communicator has %d and counted %d" % (world, world_formed, ranks_counted))
    assert B % world == 0
    B_local = B // world

    def dev_sync():
        if not cpu:
            torch.cuda.synchronize()

    if image:
        if rank == 0:
            # MIOpen's per-shape search (--miopen-benchmark, or a find mode other than FAST in the environment) runs inside
            # the first step and says nothing for minutes: a line a minute on stderr tells whoever watches the run (and
            # gpurun's silence limit) that it is alive
            import threading

            def heartbeat():
                t_start = time.perf_counter()
                while True:
                    time.sleep(60.0)
                    print("[bench] alive, %.0f s" % (time.perf_counter() - t_start), file=sys.stderr, flush=True)
            threading.Thread(target=heartbeat, daemon=True).start()
        net = build_image_net(channels, N_train)
        torch.backends.cudnn.benchmark = bool(args.miopen_benchmark)
        g = torch.Generator().manual_seed(rank)
        x = torch.randn(B_local, channels, side, side, generator=g).cuda()
        if args.channels_last:
            net.model.to(memory_format=torch.channels_last)
            x = x.contiguous(memory_format=torch.channels_last)
        y = torch.randint(0, 10, (B_local,), generator=g).cuda()
    else:
        net = build_net(din, W, L, N_train, lr, momentum, wd, cuda=not cpu)
        x, y = shard_data(B_local, din, rank, device)
    spec = None
    if dist_on:
        from dun_amd.distributed import shard
        spec = shard(net, sync_bn=not args.no_sync_bn)
        spec.force_collectives = bool(args.rehearse_dist)

    def barrier():
        dev_sync()
        if dist_on:
            dist.barrier()
        dev_sync()

    def dist_fields():
        return {"world_size_formed": world_formed, "ranks_counted_by_allreduce": ranks_counted,
                "launcher": "self (bench.py started the ranks)" if os.environ.get("DUN_BENCH_SELF_LAUNCHED") else
                            ("torch.distributed.run / external" if world > 1 else "single process"),
                "backend": (dist.get_backend() if dist_on else None)}

    if args.predict:
        step = (lambda: net.predict(x, get_std=True)) if not image else (lambda: net.predict(x))
        net.fit(x, y)  # one training step so that the BatchNorm running statistics and q are not the initial ones
        for _ in range(args.warmup):
            step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = step()
        barrier()
        dt = time.perf_counter() - t0
        if dist_on:
            t = torch.tensor([dt], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        first = out[0] if isinstance(out, tuple) else out
        if not bool(torch.isfinite(first).all()):
            raise SystemExit("bench.py: non-finite predictions")
        if rank == 0:
            line = {
                "metric": "DUN predict samples/sec", "value": B * args.steps / dt, "unit": "samples/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
                "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": "%s: predict, global batch %d (%d rows/GPU)" % (args.workload, B, B_local),
                           "parallelism": "dp%d batch-sharded (no collective: eval-mode BatchNorm)" % world,
                           "distributed": dist_fields()}}
            if not cpu:
                label_exploratory(line)
            print(json.dumps(line), file=json_out, flush=True)
        if dist_on:
            dist.barrier()
            dist.destroy_process_group()
        return

    last = None
    in_step = None
    for i in range(args.warmup):
        last = net.fit(x, y)
        if i == 0 and not cpu:
            # The first device-wide synchronise of a process that has work behind it is followed by ONE slow step (+60-70 ms
            # at the wide configuration: 609, 667, 600, 600, ... ms per step with a synchronise between steps; 609, 599, 599, ...
            # without; scripts/dev/step_mode.py) -- runtime set-up, not throughput.  Taken here, inside the warm-up, so that
            # the barrier in front of the timed steps is not the first one.
            dev_sync()
        if image and rank == 0:
            dev_sync()
            print("[bench] warm-up step %d done" % i, file=sys.stderr, flush=True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = net.fit(x, y)
    barrier()
    dt = time.perf_counter() - t0
    if dist_on:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if not all(math.isfinite(v) for v in last):
        raise SystemExit("bench.py: non-finite training metrics %r" % (last,))
    if not image and not cpu and not args.no_kernel_rooflines and not small_net_path(B_local, W):
        # one more step, outside the timed region and behind it (clocks settled), with every GEMM launch of rank 0
        # bracketed by HIP events: the figure the roofline quotes.  Every rank runs the step (same collectives).
        if rank == 0:
            in_step = gemm_in_step(net, x, y, B_local)
        else:
            net.fit(x, y)
        barrier()

    result = None
    if rank == 0 and image:
        ms_step = dt / args.steps * 1e3
        result = {
            "metric": "DUN ELBO train samples/sec", "value": B * args.steps / dt, "unit": "samples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: classification DUN_VI.fit, ResNet-50 DUN (start_depth 1, end_depth 13, D=12, "
                                   "10 classes) on %dx%dx%d inputs, global batch %d (%d/GPU), conv trunk on MIOpen, "
                                   "head + CE/ELBO on the HIP kernels" % (args.workload, channels, side, side, B, B_local),
                       "parallelism": "dp%d batch-sharded, %s BatchNorm2d" % (
                           world, "single-device" if not dist_on else
                           "synchronised" if (not args.no_sync_bn and spec is not None and spec.sync_bn2d_modules > 0)
                           else "per-replica"),
                       "distributed": dist_fields(), "train_metrics_last_step": list(last)}}
        if not args.no_kernel_rooflines:
            kr = image_head_rooflines(net, B_local)
            result["roofline_other_kernels"] = {
                k: {"bound": "hbm", "achieved": v["gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": v["gbs"] / HBM_PEAK_GBS, "ms_per_launch": v["ms"]} for k, v in kr.items()}
        print(json.dumps(result), file=json_out, flush=True)
    if image:
        if dist_on:
            dist.barrier()
            dist.destroy_process_group()
        return
    if rank == 0:
        ms_step = dt / args.steps * 1e3
        value = B * args.steps / dt
        flop_step = 6.0 * B * (L * W * W + din * W + (L + 1) * W * 1)
        result = {
            "metric": "DUN ELBO train samples/sec", "value": value, "unit": "samples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: regression DUN_VI.fit, residual MLP %d blocks x width %d, din %d, global batch "
                                   "%d (%d rows/GPU), N_train %d, SGD(lr %g, momentum %g)" %
                                   (args.workload, L, W, din, B, B_local, N_train, lr, momentum),
                       "parallelism": "dp%d batch-sharded, %s BatchNorm" % (world, "per-replica" if args.no_sync_bn and dist_on else "synchronised" if dist_on else "single-device"),
                       "distributed": dist_fields(),
                       "step_tflop": flop_step / 1e12, "train_metrics_last_step": list(last)},
            "step_tflops_per_gpu": flop_step / world / (ms_step / 1e3) / 1e12,
        }
        if cpu:  # never to be mistaken for a measurement
            result["metric"] = "REHEARSAL ONLY (CPU ranks, gloo, tests/cpu_backend.py stand-in): not a DUN_AMD measurement"
            result["data"] = "synthetic, tiny; CPU rehearsal of the launch path"
            result["dtype"] = "f32 (torch CPU stand-in)"
    if rank == 0 and not args.no_kernel_rooflines:
        kr = kernel_rooflines(net, B_local, reps=5 if args.workload == "c3_wide" else 20, y=y)
        # dominant kernel of the step: the trunk GEMM (3*L launches per step); quote the forward instantiation
        dom = kr["gemm_fwd"]
        small = small_net_path(B_local, W)
        result["roofline"] = {"bound": "mfma", "kernel": dom["kernel"], "achieved": dom["tflops"],
                              "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s", "frac": dom["tflops"] / MFMA_F32_PEAK_TF,
                              "traffic": pmc_traffic("gemm_fwd", args.workload, world), "ms_per_launch": dom["ms"],
                              "flop_per_launch": 2.0 * B_local * W * W,
                              "timing": "back-to-back launches of the kernel on the step's buffers (HIP events)"}
        if result["roofline"]["traffic"] is not None:
            result["roofline"]["traffic_source"] = ("profiles/pmc_traffic.json: rocprofv3 PMC passes of an earlier run of this "
                                                    "workload (2*FETCH_SIZE + WRITE_SIZE per launch), NOT measured by this run")
        if in_step and "fwd" in in_step:
            # the figure the step is made of: the same kernel timed inside a real training step (a warm-up step of this
            # run, every GEMM launch bracketed by HIP events); the back-to-back figure stays beside it
            live = dict(achieved=result["roofline"]["achieved"], frac=result["roofline"]["frac"],
                        ms_per_launch=result["roofline"]["ms_per_launch"])
            f = in_step["fwd"]
            result["roofline"].update(achieved=f["tflops"], frac=f["frac"], ms_per_launch=f["ms_per_launch"],
                                      timing="inside a real training step of this run (one extra step behind the timed ones): HIP events "
                                             "around each of the %d forward GEMM launches (dun_gemm_profile_begin/_end)" % f["launches"],
                                      back_to_back=live, in_step=in_step)
        if small:  # latency-bound regime: the figure is whole-forward flops, not a roofline claim
            result["roofline"]["flop_per_launch"] = dom["tflops"] * dom["ms"] * 1e9
            result["roofline"]["note"] = "small net: launch/latency bound, fp32 vector ALUs; fraction of the f32 matrix peak shown for scale only"
        result["roofline_other_kernels"] = {
            k: ({"bound": "mfma", "achieved": v["tflops"], "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                 "frac": v["tflops"] / MFMA_F32_PEAK_TF, "ms_per_launch": v["ms"]} if "tflops" in v else
                {"bound": "hbm", "achieved": v["gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": v["gbs"] / HBM_PEAK_GBS, "ms_per_launch": v["ms"]})
            for k, v in kr.items() if k != "gemm_fwd"}
    if rank == 0 and not cpu:
        label_exploratory(result)
    if rank == 0 and world == 1 and not cpu and args.workload == "c3_wide" and not args.no_secondary:
        result["secondary"] = secondary_c2()
    if rank == 0 and not args.no_cpu_baseline and world == 1:
        # free the GPU-side buffers' host shadows are irrelevant; the oracle runs on the host cores only
        B_cpu = args.cpu_batch or (2048 if args.workload == "c3_wide" else B)
        result["cpu_baseline"] = cpu_baseline(din, W, L, N_train, lr, momentum, wd, B_cpu)
    if rank == 0:
        print(json.dumps(result), file=json_out, flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
