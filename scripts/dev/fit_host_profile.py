"""Where the host time of a graphed DUN_VI.fit goes (UCI shape): cProfile over 2000 steps."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

din, W, L, B, N, lr, mom, wd = bench.WORKLOADS["c2_uci"]
net = bench.build_net(din, W, L, N, lr, mom, wd)
x, y = bench.shard_data(B, din, 0, torch.device("cuda", 0))
for _ in range(20):
    net.fit(x, y)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2000):
    net.fit(x, y)
torch.cuda.synchronize()
print("ms/step %.4f" % ((time.perf_counter() - t0) / 2000 * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(2000):
    net.fit(x, y)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
