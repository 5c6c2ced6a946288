"""Where a workgroup of the four-stage GEMM spends its time (diagnostic build, -DDUN_GEMM_TIMING): shader-clock stamps at
workgroup start, first MFMA, end of the K loop and workgroup end, per workgroup, plus the hardware slot it ran on.

    python scripts/gemm_phases.py --build      (CPU container: writes dun_amd/libdun_hip_timing.so)
    python scripts/gemm_phases.py              (GPU box: runs the C3 forward / dX / dW GEMMs once each with stamps)
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
TAG = os.environ.get("DUN_TIMING_TAG", "")  # e.g. DUN_TIMING_TAG=nt DUN_TIMING_FLAGS=-DDUN_EPI_STORE_AUX=2 with --build
LIB = os.path.join(ROOT, "dun_amd", "libdun_hip_timing%s.so" % TAG)

if "--build" in sys.argv:
    from dun_amd import build as B
    B.build_library()
    objdir = os.path.join(ROOT, "dun_amd", "build")
    o = os.path.join(objdir, "gemm_timing%s.o" % TAG)
    subprocess.run([B._hipcc(), "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-DDUN_GEMM_TIMING"] +
                   os.environ.get("DUN_TIMING_FLAGS", "").split() + ["-c",
                    os.path.join(B.CSRC, "gemm.hip"), "-o", o], check=True)
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in B.SOURCES if s != "gemm.hip"] + [o]
    subprocess.run([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs, check=True)
    print("built", LIB)
    sys.exit(0)

os.environ["DUN_HIP_LIB"] = LIB
import ctypes
import numpy as np
import torch
from dun_amd import kernels, _lib as L

be = kernels.backend()
lib = L.lib()
lib.dun_debug_gemm_timing.argtypes = [ctypes.c_void_p]
M, N, K = 65536, 2048, 2048
g = torch.Generator(device="cuda").manual_seed(0)
A = torch.randn(M, K, device="cuda", generator=g)
Wm = torch.randn(N, K, device="cuda", generator=g) * 0.02
bias = torch.randn(N, device="cuda", generator=g)
R = torch.randn(M, N, device="cuda", generator=g)
out = torch.empty(M, N, device="cuda")
skws = torch.empty(2 * N * N, device="cuda")
gw = torch.empty(N, N, device="cuda")
stamps = torch.zeros(8192 * 2 * 8, dtype=torch.int64, device="cuda")


def run(name, fn, nwg):
    for _ in range(12):
        fn()  # settle the clocks
    torch.cuda.synchronize()
    stamps.zero_()
    assert lib.dun_debug_gemm_timing(stamps.data_ptr()) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record()
    torch.cuda.synchronize()
    assert lib.dun_debug_gemm_timing(None) == 0
    t = stamps.cpu().numpy().reshape(-1, 8)[:nwg].astype(np.int64)
    ms = e0.elapsed_time(e1)
    # the counters of different XCDs are not aligned with each other: only differences inside a workgroup (and inside
    # a CU) mean anything.  Tick length: two workgroups at a time fill a CU for the whole launch.
    ncu = 256
    tick_ns = ms * 1e6 / ((t[:, 3] - t[:, 0]).sum() / (2.0 * ncu))
    pro, loop, epi, life = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 3] - t[:, 0]
    print("%s: %.3f ms, %d workgroups, clock tick %.2f ns" % (name, ms, nwg, tick_ns))
    issue = t[:, 4] - t[:, 2]
    for lab, v in (("start -> first MFMA", pro), ("K loop", loop), ("epilogue, to last instruction", issue),
                   ("epilogue, to stores drained", epi), ("lifetime", life)):
        print("    %-30s mean %8.2f us  median %8.2f  p95 %8.2f   %.1f %% of lifetime" % (
            lab, v.mean() * tick_ns / 1e3, np.median(v) * tick_ns / 1e3, np.percentile(v, 95) * tick_ns / 1e3,
            100.0 * v.sum() / life.sum()))
    # gap between consecutive workgroups of one hardware slot: key = (xcc, se, cu, wave slot of wave 0)
    hw, xcc = t[:, 5], t[:, 6] & 0xF
    key = (xcc << 32) | (hw & 0xFFFFF)  # wave_id, simd, pipe, cu, sh, se, tg_id -- slot of the first wave
    cu = (xcc << 16) | ((hw >> 8) & 0xFF) | (((hw >> 13) & 7) << 8)
    gaps = []
    for c in np.unique(cu):
        idx = np.where(cu == c)[0]
        # two workgroups run at a time on a CU: order by start; the successor of a workgroup is the first one that
        # starts after it ends
        st, en = np.sort(t[idx, 0]), np.sort(t[idx, 3])
        for e in en:
            nxt = st[st >= e]
            if len(nxt):
                gaps.append(nxt[0] - e)
    gaps = np.array(gaps if gaps else [0])
    print("    CUs seen %d; workgroups per CU %.1f; end of a workgroup -> start of the next on the same CU: mean %.2f us  median %.2f  p95 %.2f" % (
        len(np.unique(cu)), nwg / len(np.unique(cu)), gaps.mean() * tick_ns / 1e3, np.median(gaps) * tick_ns / 1e3,
        np.percentile(gaps, 95) * tick_ns / 1e3))
    # are the workgroups in phase?  per XCD (its counters are one clock): how many of its 64 resident workgroups sit in
    # their epilogue at the same moment (64 * epilogue / lifetime ~ 2-3 if phases were uniform)
    conc, conc_first = [], []
    for x in np.unique(xcc):
        idx = np.where(xcc == x)[0]
        ev = sorted([(a, 1) for a in t[idx, 2]] + [(b, -1) for b in t[idx, 4]])
        cur, series = 0, []
        for _, d in ev:
            cur += d
            series.append(cur)
        conc.append(max(series))
        order = idx[np.argsort(t[idx, 0])]
        per_round = len(idx) // max(1, len(idx) // 64) if len(idx) >= 64 else len(idx)
        # spread of the start times of the LAST 64 workgroups started on this XCD (= the last round), in units of a lifetime
        last = np.sort(t[idx, 0])[-64:] if len(idx) >= 128 else np.sort(t[idx, 0])
        conc_first.append((last.max() - last.min()) / max(1.0, life.mean()))
    print("    in phase?  workgroups of one XCD inside their epilogue at once: max %s of 64 resident (uniform phases would give ~%.1f); "
          "start-time spread of an XCD's last 64 workgroups: %.2f lifetimes" % (
              sorted(conc)[len(conc) // 2], 64.0 * issue.mean() / life.mean(), float(np.median(conc_first))))
    print("    loop share of the launch: %.3f; ideal loop time at 157.3 TF: %.2f us per workgroup (two per CU)" % (
        loop.sum() / life.sum(), 2.0 * 2 * 128 * 128 * K / (157.3e12 / 256) * 1e6 / (2 if nwg == 512 else 1)))
    sys.stdout.flush()


run("forward (bias + ReLU + BN column statistics)",
    lambda: be.gemm(A, Wm, M, N, K, bias=bias, relu=True, colstats=True, out=out), 8192)
run("plain (no epilogue work)", lambda: be.gemm(A, Wm, M, N, K, out=out), 8192)
u1 = torch.randn(M, 1, device="cuda", generator=g) * 1e-3
v1 = torch.randn(1, N, device="cuda", generator=g)
mean, rstd = torch.randn(N, device="cuda", generator=g), torch.rand(N, device="cuda", generator=g) + 0.5
run("dX (residual seed, rank-1 term, BN-backward statistics)",
    lambda: be.gemm(A, Wm, M, N, K, R=R, U=u1, V=v1, out=out, bn=(A, mean, rstd)), 8192)
run("dW (split-K 2)", lambda: be.gemm(A, R, N, N, M, a_kcontig=False, b_kcontig=False, out=gw, splitk_ws=skws), 512)
