"""world_size-2 gloo test of the batch-sharded path (CPU stand-in backend): two ranks, each on half the rows of a
global batch, must reproduce the single-process step on the full batch -- objective, metrics, parameters, BatchNorm
running statistics -- with synchronised BatchNorm; and must differ with per-replica statistics."""
import contextlib
import io
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.util import Case


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _net(case):
    from dun_amd import kernels
    from tests.cpu_backend import CpuBackend
    from tests.test_host_cpu import build_cpu_net
    kernels.set_backend(CpuBackend())
    return build_cpu_net(case)


def _state(net):
    out = {"flat": net.model.flat_params.clone(), "rm": net.model.running_mean.clone(),
           "rv": net.model.running_var.clone()}
    if hasattr(net.prob_model, "q_logits"):
        out["q"] = net.prob_model.q_logits.data.clone()
    if net.regression:
        out["ls"] = net.f_neg_loglike.log_std.data.clone()
    return out


class _Loader(list):
    """Two batches of a tensor dataset, with the `.dataset` the sweeps ask the length of."""
    def __init__(self, x, y, n_dataset):
        h = x.shape[0] // 2
        super(_Loader, self).__init__([(x[:h], y[:h]), (x[h:2 * h], y[h:2 * h])])
        self.dataset = range(n_dataset)


def _sweeps(net, x, y, n_dataset):
    """get_exact_d_posterior (+ get_exact_ELBO for VI) over a two-batch loader, eval-mode BatchNorm."""
    loader = _Loader(x, y, n_dataset)
    post, log_marg = net.get_exact_d_posterior(loader, train_bn=False, logposterior=True)
    out = [post.cpu().numpy(), log_marg]
    if hasattr(net, "get_exact_ELBO"):
        out.append(net.get_exact_ELBO(loader, train_bn=False))
    return out


def _record_collectives(log):
    """Wrap torch.distributed's collectives in this process: every call appends (op, shape, dtype, async, reduce op)
    to `log` before it is issued.  Two ranks whose logs differ would wait in different collectives -- the one
    product-side way to get "both ranks sit in all_reduce with every kernel done"."""
    real = {k: getattr(dist, k) for k in ("all_reduce", "broadcast", "barrier")}

    def all_reduce(t, op=dist.ReduceOp.SUM, group=None, async_op=False):
        log.append(("all_reduce", tuple(t.shape), str(t.dtype), bool(async_op), str(op)))
        return real["all_reduce"](t, op=op, group=group, async_op=async_op)

    def broadcast(t, src=0, group=None, async_op=False):
        log.append(("broadcast", tuple(t.shape), str(t.dtype), bool(async_op), int(src)))
        return real["broadcast"](t, src=src, group=group, async_op=async_op)

    def barrier(*a, **k):
        log.append(("barrier",))
        return real["barrier"](*a, **k)

    dist.all_reduce, dist.broadcast, dist.barrier = all_reduce, broadcast, barrier


def _worker(rank, world, port, name, sync_bn, steps, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    log = []
    _record_collectives(log)
    try:
        from dun_amd.distributed import shard
        c = Case(name)
        net = _net(c)
        if rank == 1:  # replicas may start different: shard() must broadcast rank 0's state
            net.model.flat_params.add_(0.5)
        with contextlib.redirect_stdout(io.StringIO()):
            shard(net, sync_bn=sync_bn)
        x, y = c.t("x"), c.y()
        B = x.shape[0] // world
        xs, ys = x[rank * B:(rank + 1) * B], y[rank * B:(rank + 1) * B]
        rets = [net.fit(xs, ys) for _ in range(steps)]
        ev = net.eval(xs, ys)
        st = _state(net)
        sweep = _sweeps(net, xs, ys, c.cfg["N"] // world)
        ret[rank] = (rets, ev, {k: v.numpy() for k, v in st.items()}, sweep, log)
    finally:
        dist.destroy_process_group()


def _run(name, sync_bn, steps=2):
    mgr = mp.Manager()
    ret = mgr.dict()
    port = _free_port()
    mp.spawn(_worker, args=(2, port, name, sync_bn, steps, ret), nprocs=2, join=True)
    return ret[0], ret[1]


def _single(name, steps=2):
    c = Case(name)
    net = _net(c)
    x, y = c.t("x"), c.y()
    rets = [net.fit(x, y) for _ in range(steps)]
    ev = net.eval(x, y)
    # the same two global batches the two ranks see half of each: rows [0,h/2)+[B/2,B/2+h/2), ...
    B = x.shape[0]
    h = B // 4
    idx = torch.cat([torch.arange(0, h), torch.arange(B // 2, B // 2 + h), torch.arange(h, 2 * h),
                     torch.arange(B // 2 + h, B // 2 + 2 * h)])
    sweep = _sweeps(net, x[idx], y[idx], c.cfg["N"])
    from dun_amd import kernels
    kernels.set_backend(None)
    return rets, ev, {k: v.numpy() for k, v in _state(net).items()}, sweep


@pytest.mark.parametrize("name", ["c3_reduced", "cls_fc", "mll_small", "prior_gauss", "prior_mll"])
def test_two_ranks_match_single_process(name):
    r0, r1 = _run(name, sync_bn=True)
    rets, ev, st, single_sweep = _single(name)
    for (a, b, s) in zip(r0[0], r1[0], rets):
        assert np.allclose(a, b, rtol=1e-6, equal_nan=True)          # both ranks report the global-batch metrics
        assert np.allclose(a, s, rtol=2e-5, atol=1e-6, equal_nan=True), (a, s)
    assert np.allclose(r0[1], ev, rtol=2e-5, equal_nan=True)
    # full-dataset sweeps: each rank walks its half of every global batch, the per-depth sums are all-reduced
    sw = single_sweep
    for a, b, s_ in zip(r0[3], r1[3], sw):
        assert np.allclose(a, b, rtol=1e-6)
        assert np.allclose(a, s_, rtol=2e-5, atol=1e-5), (a, s_)
    # both ranks issued the same collectives in the same order (op, shape, dtype, sync/async): fit x2, eval, sweeps
    assert len(r0[4]) > 0 and r0[4] == r1[4], [(i, a, b) for i, (a, b) in enumerate(zip(r0[4], r1[4])) if a != b][:3]
    for k in st:
        assert np.allclose(r0[2][k], r1[2][k], rtol=0, atol=0), k        # replicas stay bit-identical
        scale = np.abs(st[k]).max()
        assert np.abs(r0[2][k] - st[k]).max() <= 2e-5 * scale + 1e-7, k


def test_per_replica_batchnorm_differs():
    r0, _ = _run("c3_reduced", sync_bn=False, steps=1)
    rets, _, st, _ = _single("c3_reduced", steps=1)
    # per-replica statistics are a different (DataParallel-like) computation: close, but not the same numbers
    assert abs(r0[0][0][0] - rets[0][0]) > 1e-6 * abs(rets[0][0])
    assert np.abs(r0[2]["rm"] - st["rm"]).max() > 0


# ---- the shapes an 8-GPU run meets: more ranks, ragged gradient buckets, every switch of shard() ----------------------
def _syn_net(cfg):
    """A seeded synthetic DUN (no fixture): din 6, width 32, cfg['L'] blocks, regression, VI or exact marginal."""
    from dun_amd import kernels
    from dun_amd.probability import depth_categorical, depth_categorical_VI
    from dun_amd.stochastic_fc_models import arq_uncert_fc_resnet
    from dun_amd.training_wrappers import DUN, DUN_VI
    from tests.cpu_backend import CpuBackend
    kernels.set_backend(CpuBackend())
    torch.manual_seed(7)
    L = cfg["L"]
    model = arq_uncert_fc_resnet(6, 1, 32, L)
    pm = (depth_categorical_VI if cfg["vi"] else depth_categorical)([1.0 / (L + 1)] * (L + 1), cuda=False)
    with contextlib.redirect_stdout(io.StringIO()):
        net = (DUN_VI if cfg["vi"] else DUN)(model, pm, 96, lr=0.05, momentum=0.9, weight_decay=1e-4, cuda=False,
                                             regression=True)
    g = torch.Generator().manual_seed(8)
    x = torch.randn(96, 6, generator=g)      # 96 rows: 48 / 32 / 24 per rank
    y = torch.tanh(x[:, :1] - 0.5 * x[:, 1:2]) + 0.1 * torch.randn(96, 1, generator=g)
    return net, x, y


def _syn_worker(rank, world, port, cfg, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    log = []
    _record_collectives(log)
    try:
        from dun_amd.distributed import shard
        net, x, y = _syn_net(cfg)
        if rank != 0:
            net.model.flat_params.add_(0.25 * rank)      # shard() must make the replicas identical
        with contextlib.redirect_stdout(io.StringIO()):
            spec = shard(net, sync_bn=cfg["sync_bn"], overlap=cfg["overlap"], grad_bucket_blocks=cfg["buckets"])
        assert spec.world == world and spec.grad_bucket_blocks == cfg["buckets"]
        B = x.shape[0] // world
        xs, ys = x[rank * B:(rank + 1) * B], y[rank * B:(rank + 1) * B]
        rets = [net.fit(xs, ys) for _ in range(3)]
        ret[rank] = (rets, {k: v.numpy() for k, v in _state(net).items()}, log)
    finally:
        dist.destroy_process_group()


def _syn_single(cfg):
    from dun_amd import kernels
    try:
        net, x, y = _syn_net(cfg)
        rets = [net.fit(x, y) for _ in range(3)]
        return rets, {k: v.numpy() for k, v in _state(net).items()}
    finally:
        kernels.set_backend(None)


@pytest.mark.parametrize("world,L,buckets,overlap,vi", [
    (4, 5, 1, True, True),      # one message per block
    (4, 5, 2, True, True),      # 5 blocks in buckets of two: a ragged bucket
    (4, 5, 3, True, True),      # 3 + 2
    (2, 5, 3, False, True),     # blocking gradient all-reduces
    (4, 5, 2, True, False),     # exact marginal likelihood (DUN): the per-depth sums gate the backward
    (3, 4, 8, False, False),    # a bucket larger than the net, an odd world size
])
def test_more_ranks_and_ragged_gradient_buckets_match_single_process(world, L, buckets, overlap, vi):
    cfg = dict(L=L, buckets=buckets, overlap=overlap, vi=vi, sync_bn=True)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_syn_worker, args=(world, _free_port(), cfg, ret), nprocs=world, join=True)
    rets, st = _syn_single(cfg)
    for r in range(world):
        for a, s_ in zip(ret[r][0], rets):
            assert np.allclose(a, s_, rtol=2e-5, atol=1e-6, equal_nan=True), (r, a, s_)
        for k in st:
            assert np.array_equal(ret[r][1][k], ret[0][1][k]), (r, k)            # replicas bit-identical
            scale = np.abs(st[k]).max()
            assert np.abs(ret[r][1][k] - st[k]).max() <= 2e-5 * scale + 1e-7, (r, k)
        assert ret[r][2] == ret[0][2], [(i, a, b) for i, (a, b) in enumerate(zip(ret[r][2], ret[0][2])) if a != b][:3]
    # the gradient messages follow the bucket size: block weights travel as ceil(L / buckets) messages per step
    log = ret[0][2]
    w_msgs = [e for e in log if e[0] == "all_reduce" and e[2] == "torch.float32" and len(e[1]) == 3]
    per_step = [e[1][0] for e in w_msgs[:len(w_msgs) // 3]]
    full, rest = divmod(L, buckets)
    assert per_step == [buckets] * full + ([rest] if rest else []), per_step       # deepest blocks first, ragged last bucket
    assert len(w_msgs) == 3 * len(per_step) and all(e[1][1:] == (32, 32) for e in w_msgs)
    # everything else of the flat gradient buffer (the `exclude=` slicing around the block weights): 1-D pieces
    flat_msgs = [e for e in log if e[0] == "all_reduce" and e[2] == "torch.float32" and len(e[1]) == 1 and e[1][0] > 32]
    assert len(flat_msgs) >= 3
    assert all(e[3] == overlap for e in w_msgs + flat_msgs)


def test_four_ranks_per_replica_batchnorm():
    cfg = dict(L=5, buckets=2, overlap=True, vi=True, sync_bn=False)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_syn_worker, args=(4, _free_port(), cfg, ret), nprocs=4, join=True)
    rets, st = _syn_single(cfg)
    for r in range(1, 4):
        assert np.array_equal(ret[r][1]["flat"], ret[0][1]["flat"])               # parameters: summed gradients, identical
        assert np.array_equal(ret[r][1]["q"], ret[0][1]["q"])
        assert not np.array_equal(ret[r][1]["rm"], ret[0][1]["rm"])               # statistics of each rank's own rows
        assert ret[r][2] == ret[0][2]
    assert not any(e[0] == "all_reduce" and e[1] == (2, 32) for e in ret[0][2])   # no BatchNorm exchange at all
    assert np.abs(ret[0][1]["flat"] - st["flat"]).max() > 1e-6                     # a different computation


def _img_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dun_amd import kernels
        from dun_amd.distributed import shard
        from tests.cpu_backend import CpuBackend
        from tests.test_image_dun import build
        kernels.set_backend(CpuBackend())
        c = Case("img_r18")
        net, meta = build(c, cuda=False)
        if rank == 1:
            with torch.no_grad():
                for p in net.model.parameters():
                    p.add_(0.1)
        with contextlib.redirect_stdout(io.StringIO()):
            shard(net)
        x, y = c.t("x"), c.t("y").long()
        B = x.shape[0] // world
        xs, ys = x[rank * B:(rank + 1) * B], y[rank * B:(rank + 1) * B]
        # this rank's own gradient of the global-batch loss, before the exchange
        net.model.train()
        act = net.model.forward(xs)
        prior = net.model.get_w_prior_loglike()
        out = net.prob_model.fused_objective(prior, act, ys, net.f_neg_loglike, meta["N"])
        local = torch.autograd.grad(act, net.model.output_block.weight, out.dact.view_as(act), retain_graph=False)[0]
        rets = [net.fit(xs, ys) for _ in range(2)]
        flat = torch.cat([p.data.reshape(-1) for p in net.model.parameters()])
        ret[rank] = (rets, flat.numpy(), net.prob_model.q_logits.data.numpy().copy(), local.numpy())
    finally:
        dist.destroy_process_group()


def _img_state(net):
    sd = net.model.state_dict()
    return {"params": torch.cat([p.data.reshape(-1) for p in net.model.parameters()]).numpy(),
            "running_mean": torch.cat([v.reshape(-1) for k, v in sd.items() if k.endswith("running_mean")]).numpy(),
            "running_var": torch.cat([v.reshape(-1) for k, v in sd.items() if k.endswith("running_var")]).numpy(),
            "num_batches_tracked": torch.stack([v for k, v in sd.items() if k.endswith("num_batches_tracked")]).numpy(),
            "q_logits": net.prob_model.q_logits.data.numpy().copy()}


def _img_sync_worker(rank, world, port, sync_bn, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    log = []
    _record_collectives(log)
    try:
        from dun_amd import kernels
        from dun_amd.distributed import shard
        from dun_amd.sync_bn import SyncBatchNorm2d
        from tests.cpu_backend import CpuBackend
        from tests.test_image_dun import build
        kernels.set_backend(CpuBackend())
        c = Case("img_r18")
        net, meta = build(c, cuda=False)
        keys = list(net.model.state_dict().keys())
        with contextlib.redirect_stdout(io.StringIO()):
            spec = shard(net, sync_bn=sync_bn)
        n_bn = sum(isinstance(m, torch.nn.BatchNorm2d) for m in net.model.modules())
        assert spec.sync_bn2d_modules == (n_bn if sync_bn else 0) and n_bn > 0
        assert all(isinstance(m, SyncBatchNorm2d) == sync_bn for m in net.model.modules() if isinstance(m, torch.nn.BatchNorm2d))
        assert list(net.model.state_dict().keys()) == keys       # re-classed in place: checkpoints stay interchangeable
        x, y = c.t("x"), c.t("y").long()
        B = x.shape[0] // world
        xs, ys = x[rank * B:(rank + 1) * B], y[rank * B:(rank + 1) * B]
        rets = [net.fit(xs, ys) for _ in range(2)]
        ev = net.eval(xs, ys)
        ret[rank] = (rets, ev, _img_state(net), log)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_image_dun_two_ranks_match_single_process(world):
    """Image DUN under shard(sync_bn=True): every BatchNorm2d call of the conv trunk -- blocks, adapt stacks and the
    tail each exit runs again -- normalises with the statistics of the global batch (dun_amd/sync_bn.py), so ranks on
    slices of the img_r18 batch (6 rows: 3 + 3, or 2 + 2 + 2) reproduce the single-process steps on all 6 rows."""
    from dun_amd import kernels
    from tests.cpu_backend import CpuBackend
    from tests.test_image_dun import build
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_img_sync_worker, args=(world, _free_port(), True, ret), nprocs=world, join=True)
    kernels.set_backend(CpuBackend())
    try:
        c = Case("img_r18")
        net, _ = build(c, cuda=False)
        x, y = c.t("x"), c.t("y").long()
        rets = [net.fit(x, y) for _ in range(2)]
        ev = net.eval(x, y)
        st = _img_state(net)
    finally:
        kernels.set_backend(None)
    for r in range(world):
        for a, s in zip(ret[r][0], rets):
            assert np.allclose(a, s, rtol=2e-5, atol=1e-7), (r, a, s)      # objective, nll, err of the GLOBAL batch
        # eval runs on the rank's own rows with the (synchronised) running statistics: compare what is comparable
        assert np.all(np.isfinite(ret[r][1]))
        for k, v in st.items():
            scale = np.abs(v).max()
            assert np.abs(ret[r][2][k] - v).max() <= 2e-5 * scale + 1e-7, (r, k)
            assert np.array_equal(ret[r][2][k], ret[0][2][k]), k             # replicas stay bit-identical
        assert ret[r][3] == ret[0][3] and len(ret[r][3]) > 0                 # same collectives, same order
    # the per-call exchanges are there: fp64 [2,C] messages, forward and backward
    kinds = {(e[1], e[2]) for e in ret[0][3] if e[0] == "all_reduce"}
    assert ((2, 64), "torch.float64") in kinds and ((2, 512), "torch.float64") in kinds


def test_image_dun_per_replica_batchnorm_differs():
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_img_sync_worker, args=(2, _free_port(), False, ret), nprocs=2, join=True)
    ret2 = mgr.dict()
    mp.spawn(_img_sync_worker, args=(2, _free_port(), True, ret2), nprocs=2, join=True)
    assert np.array_equal(ret[0][2]["params"], ret[1][2]["params"])          # gradients are still summed
    assert not np.array_equal(ret[0][2]["running_mean"], ret[1][2]["running_mean"])   # statistics of different rows
    assert abs(ret[0][0][0][0] - ret2[0][0][0][0]) > 1e-6 * abs(ret2[0][0][0][0])  # a different computation


def test_sync_batchnorm2d_gradients_match_full_batch_autograd():
    """The Function alone, one rank with collectives forced (a one-rank all-reduce is the identity): forward, running
    statistics and all three gradients against nn.BatchNorm2d's own autograd on the same input in fp64."""
    port = _free_port()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        from dun_amd.distributed import Shard
        from dun_amd.sync_bn import convert
        spec = Shard()
        spec.force_collectives = True
        g = torch.Generator().manual_seed(3)
        for dtype, tol in ((torch.float64, 1e-12), (torch.float32, 2e-5)):
            ref = torch.nn.BatchNorm2d(5).to(dtype)
            with torch.no_grad():
                ref.weight.copy_(torch.randn(5, generator=g)); ref.bias.copy_(torch.randn(5, generator=g))
            mine = torch.nn.Sequential(torch.nn.BatchNorm2d(5).to(dtype))
            mine[0].load_state_dict(ref.state_dict())
            assert convert(mine, spec) == 1
            x = (3.0 + 2.0 * torch.randn(4, 5, 3, 2, generator=g)).to(dtype)
            dy = torch.randn(4, 5, 3, 2, generator=g).to(dtype)
            xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
            ya, yb = ref(xa), mine(xb)
            ya.backward(dy); yb.backward(dy)
            for a, b in ((ya, yb), (xa.grad, xb.grad), (ref.weight.grad, mine[0].weight.grad), (ref.bias.grad, mine[0].bias.grad),
                         (ref.running_mean, mine[0].running_mean), (ref.running_var, mine[0].running_var)):
                assert float((a - b).detach().abs().max()) <= tol * max(1.0, float(a.detach().abs().max()))
            assert int(mine[0].num_batches_tracked) == 1
            import pickle
            again = pickle.loads(pickle.dumps(mine))      # (written by this test) the communicator does not travel
            assert again[0]._dun_dist is None and mine[0]._dun_dist is spec
            assert torch.equal(again[0].running_var, mine[0].running_var)
            mine[0].load_state_dict(ref.state_dict())
            mine.eval(); ref.eval()
            assert torch.equal(mine(x), ref(x))          # eval mode is nn.BatchNorm2d's own path
    finally:
        dist.destroy_process_group()


def test_image_dun_two_ranks_stay_identical():
    """Image DUN (conv trunk in torch, synchronised BatchNorm2d by default): loose-parameter gradients are SUM
    all-reduced, replicas that started different are made identical and stay so."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_img_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    r0, r1 = ret[0], ret[1]
    assert np.allclose(r0[0], r1[0], rtol=1e-6)              # both ranks report the global-batch objective / metrics
    assert np.array_equal(r0[1], r1[1]) and np.array_equal(r0[2], r1[2])   # replicas identical after two steps
    assert np.all(np.isfinite(r0[1]))
    assert not np.allclose(r0[3], r1[3])                     # (the ranks did see different rows)
