"""Where a block of the one-launch small-net forward spends its time: run with DUN_HIP_LIB=scripts/dev/libdun_hip_prof.so
(scripts/dev/build_prof_lib.sh).  Stamps are 10 ns ticks of wave 0 of each workgroup."""
import sys
import torch
sys.path.insert(0, ".")
import bench
net = bench.build_net(13, 100, 10, 455, 1e-2, 0.5, 0.0)
x, y = bench.shard_data(512, 13, 0, torch.device("cuda"))
for _ in range(20):
    net.fit(x, y)
torch.cuda.synchronize()
buf = net.model._get_buffers(512, 10, True)
P = 32
if len(sys.argv) > 1 and sys.argv[1] == "bwd":
    # the last launch that stamped was the backward: same stamp area
    raw = buf.sync.view(torch.int32)[64:64 + 2 * P * 128 * 4].view(2, P, 128, 4)[:, :, 104:, :].contiguous().view(torch.int64).view(2, P, 48)
    st = torch.cat([raw[0], raw[1]], dim=1).cpu().double()
    for pid in (0, 15, 31):
        s = st[pid]
        print("wg %d: prologue %.2f us" % (pid, (s[1] - s[90]) / 100))
        for ph in range(10):
            b = 1 + 5 * ph
            nxt = s[b + 5] if ph < 9 else s[94]
            print("  phase %d: stats %.2f  publish+gather %.2f  prefetch+dU+store %.2f  barrier %.2f  mm+update %.2f | t=%.2f"
                  % (ph, (s[b + 1] - s[b]) / 100, (s[b + 2] - s[b + 1]) / 100, (s[b + 3] - s[b + 2]) / 100,
                     (s[b + 4] - s[b + 3]) / 100, (nxt - s[b + 4]) / 100, (s[b] - s[90]) / 100))
        print("  total %.2f us" % ((s[94] - s[90]) / 100))
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "eval":
    # eval-mode forward (no exchange): block start -> normalised -> barrier
    net.model.eval()
    with torch.no_grad():
        for _ in range(5):
            net.model.forward(x)
    torch.cuda.synchronize()
    ebuf = net.model._get_buffers(512, 10, False)
    raw = ebuf.sync.view(torch.int32)[64:64 + 2 * P * 128 * 4].view(2, P, 128, 4)[:, :, 104:, :].contiguous().view(torch.int64).view(2, P, 48)
    st = torch.cat([raw[0], raw[1]], dim=1).cpu().double()
    for pid in (0, 15, 31):
        s = st[pid]
        print("wg %d: entry->coef %.2f  input layer %.2f  head0+barrier %.2f" % (pid, (s[91] - s[90]) / 100, (s[0] - s[91]) / 100, (s[1] - s[0]) / 100))
        for i in range(10):
            b = 2 + 6 * i
            print("  block %d: mm+head+norm+store %.2f  barrier %.2f  (to next block start %.2f) | t=%.2f"
                  % (i, (s[b + 4] - s[b]) / 100, (s[b + 5] - s[b + 4]) / 100,
                     ((s[b + 6] if i < 9 else s[92]) - s[b + 5]) / 100, (s[b] - s[90]) / 100))
        print("  tail %.2f  total %.2f us" % ((s[94] - s[92]) / 100, (s[94] - s[90]) / 100))
    sys.exit(0)
from dun_amd import kernels
kernels.backend().fc_forward(buf)  # (the backward stamps the same area: run the forward once more, same launch as in fit)
torch.cuda.synchronize()
raw = buf.sync.view(torch.int32)[64:64 + 2 * P * 128 * 4].view(2, P, 128, 4)[:, :, 104:, :].contiguous().view(torch.int64).view(2, P, 48)
st = torch.cat([raw[0], raw[1]], dim=1).cpu().double()  # [P, 96]
for pid in (0, 15, 31):
    s = st[pid]
    t0 = s[0]
    print("wg %d: head0 %.2f us" % (pid, (s[1] - s[0]) / 100))
    for i in range(10):
        b = 2 + 6 * i
        seg = [(s[b + k + 1] - s[b + k]) / 100 for k in range(5)]
        print("  block %d: mm+stats %.2f  publish+Zstore %.2f  gather %.2f  finalize+norm+store %.2f  barrier %.2f  head %.2f | t=%.2f"
              % (i, seg[0], seg[1], seg[2], seg[3], seg[4], (s[b + 6] - s[b + 5]) / 100 if i < 9 else float("nan"), (s[b] - t0) / 100))
for pid in (0, 15, 31):
    s = st[pid]
    print("wg %d: entry->coef %.2f  input layer %.2f | loop end->head(D) %.2f  dw partials %.2f  partial sums + finalize %.2f | total %.2f us"
          % (pid, (s[91] - s[90]) / 100, (s[0] - s[91]) / 100, (s[92] - s[2 + 6 * 9 + 5]) / 100, (s[93] - s[92]) / 100,
             (s[94] - s[93]) / 100, (s[94] - s[90]) / 100))
print("entry spread over workgroups (us):", ((st[:, 90] - st[:, 90].min()) / 100).tolist())
print("exit spread over workgroups (us):", ((st[:, 94] - st[:, 90].min()) / 100).tolist())
print("spread of block-5 gather start over workgroups (us):", ((st[:, 2 + 30 + 2] - st[:, 2 + 30 + 2].min()) / 100).tolist())
