"""Kernel launches behind one call of the wrapper API on the UCI-sized net (run under rocprofv3 --kernel-trace --stats):
argument = eval | predict | predict_mu | layer_predict | sweep"""
import sys
import torch
sys.path.insert(0, ".")
import bench
what = sys.argv[1]
din, W, L, B, N_train, lr, momentum, wd = bench.WORKLOADS["c2_uci"]
net = bench.build_net(din, W, L, N_train, lr, momentum, wd, cuda=True)
x, y = bench.shard_data(B, din, 0, torch.device("cuda", 0))
net.fit(x, y); net.fit(x, y)
fn = {"eval": lambda: net.eval(x, y), "predict": lambda: net.predict(x, get_std=True), "predict_mu": lambda: net.predict(x),
      "layer_predict": lambda: net.layer_predict(x)}[what]
for _ in range(3):
    fn()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(1000):
    fn()
torch.cuda.synchronize()
print("%s: %.1f us per call" % (what, (time.perf_counter() - t0) * 1e3), flush=True)
