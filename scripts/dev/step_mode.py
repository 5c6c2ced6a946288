"""Per-step wall time of the wide net from a cold process, with and without a synchronise between steps."""
import sys, os, time, torch
sys.path.insert(0, os.getcwd())
import bench
din, W, L, B, N_train, lr, momentum, wd = bench.WORKLOADS["c3_wide"]
net = bench.build_net(din, W, L, N_train, lr, momentum, wd)
x, y = bench.shard_data(B, din, 0, torch.device("cuda"))
sync = len(sys.argv) > 1 and sys.argv[1] == "sync"
ts = []
t_prev = time.perf_counter()
for i in range(10):
    if sync: torch.cuda.synchronize()
    net.fit(x, y)
    if sync: torch.cuda.synchronize()
    t = time.perf_counter(); ts.append((t - t_prev) * 1e3); t_prev = t
print("sync" if sync else "no sync", "per-step ms:", " ".join("%.1f" % t for t in ts), flush=True)
