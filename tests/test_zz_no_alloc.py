"""dun_elbo_gauss_static -- the entry point that allocates nothing (partial sums in a device array inside the library)
-- against dun_elbo_gauss (partial sums in the caller's workspace), through the C ABI: bit for bit, on two calls in a
row, the second with a smaller input (the array keeps the first call's contents), and the size limit."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _problem(D, B, seed):
    g = torch.Generator().manual_seed(seed)
    act = torch.randn(D, B, generator=g).cuda()
    y = torch.randn(B, generator=g).cuda()
    ql = (0.3 * torch.randn(D, generator=g)).cuda()
    prior = torch.randn(D, generator=g).cuda()
    return act, y, ql, prior


def _both(lib, L, D, B, seed, with_prior):
    act, y, ql, prior = _problem(D, B, seed)
    ls = torch.tensor([0.1], device="cuda")
    logprior = torch.full((D,), 1.0 / D, device="cuda").log()
    N, sp = float(1 << 20), L.stream_ptr(act.device)
    pr = L.f32(prior) if with_prior else None
    n_ws = lib.dun_elbo_gauss_workspace_doubles(D, B)
    ws = torch.empty(n_ws, device="cuda", dtype=torch.float64)
    res_ws = torch.full((L.res_floats(D),), float("nan"), device="cuda")
    dact_ws = torch.full_like(act, float("nan"))
    L.check(lib.dun_elbo_gauss(L.f32(act), L.f32(y), L.f32(ls), L.f32(ql), L.f32(logprior), pr, D, B, N, 1.0, 1e-35,
                               L.f32(dact_ws), L.f32(res_ws), L.ptr(ws), n_ws, sp), "dun_elbo_gauss")
    res_st = torch.full_like(res_ws, float("nan"))
    dact_st = torch.full_like(act, float("nan"))
    torch.cuda.synchronize()
    # the call takes no device memory: this process's allocator state is unchanged across it (device-wide free memory
    # would also see other processes), and it runs under stream capture, which fails on any allocation -- the use case
    # the entry exists for
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        L.check(lib.dun_elbo_gauss_static(L.f32(act), L.f32(y), L.f32(ls), L.f32(ql), L.f32(logprior), pr, D, B, N, 1.0,
                                          1e-35, L.f32(dact_st), L.f32(res_st), L.stream_ptr(act.device)),
                "dun_elbo_gauss_static (captured)")
    assert bool(torch.isnan(res_st).all())                   # captured, not run
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(res_st, res_ws) and torch.equal(dact_st, dact_ws), (D, B, "replayed")
    res_st.fill_(float("nan")); dact_st.fill_(float("nan"))
    torch.cuda.synchronize()
    alloc_before, reserved_before = torch.cuda.memory_allocated(), torch.cuda.memory_reserved()
    L.check(lib.dun_elbo_gauss_static(L.f32(act), L.f32(y), L.f32(ls), L.f32(ql), L.f32(logprior), pr, D, B, N, 1.0,
                                      1e-35, L.f32(dact_st), L.f32(res_st), sp), "dun_elbo_gauss_static")
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated() == alloc_before and torch.cuda.memory_reserved() == reserved_before
    assert bool(torch.isfinite(res_ws).all()) and bool(torch.isfinite(dact_ws).all())
    assert torch.equal(res_st, res_ws) and torch.equal(dact_st, dact_ws), (D, B)


def test_static_entry_matches_workspace_entry_on_two_calls_in_a_row():
    assert torch.cuda.is_available()
    from dun_amd import _lib as L
    lib = L.lib()
    assert lib.dun_static_init() == 0, lib.dun_last_error()
    assert lib.dun_static_init() == 0
    _both(lib, L, 51, 65536, 1, True)     # the wide configuration: 51 x 16 partial sums
    _both(lib, L, 11, 512, 2, False)      # smaller input straight after: 11 x 1 slots of the same array
    _both(lib, L, 64, 200003, 3, True)    # ragged batch (scalar-load kernel), more slots than either call before
    _both(lib, L, 1, 1, 4, False)


def test_static_entry_rejects_what_does_not_fit():
    from dun_amd import _lib as L
    lib = L.lib()
    cap = lib.dun_elbo_gauss_static_max_doubles()
    assert cap == 65536
    D, B = 51, 1 << 23                     # 51 x 2048 partial sums
    assert lib.dun_elbo_gauss_workspace_doubles(D, B) > cap
    act, y, ql, _ = _problem(D, 1 << 10, 5)   # (never dereferenced: the call is refused before any launch)
    ls = torch.zeros(1, device="cuda")
    logprior = torch.full((D,), 1.0 / D, device="cuda").log()
    res = torch.zeros(L.res_floats(D), device="cuda")
    rc = lib.dun_elbo_gauss_static(L.f32(act), L.f32(y), L.f32(ls), L.f32(ql), L.f32(logprior), None, D, B, 1.0, 1.0,
                                   1e-35, L.f32(act), L.f32(res), L.stream_ptr(act.device))
    assert rc == 1 and b"built-in scratch" in lib.dun_last_error()
    torch.cuda.synchronize()
