"""Where the host side of a captured small-net step goes (development aid)."""
import sys, time
import torch
sys.path.insert(0, ".")
import bench
from dun_amd import kernels
net = bench.build_net(13, 100, 10, 455, 1e-2, 0.5, 0.0)
x, y = bench.shard_data(512, 13, 0, torch.device("cuda"))
for _ in range(50):
    net.fit(x, y)
st = net._graph_state
be = kernels.backend()
s = torch.cuda.current_stream()
def t(fn, n=3000):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
print("fit                         %.1f us" % t(lambda: net.fit(x, y)))
def a(): st["graph"].replay(); s.synchronize()
print("replay + sync               %.1f us" % t(a))
def b(): be.copy_pair(st["x"], x, st["y"], y); st["graph"].replay(); s.synchronize()
print("copy_pair + replay + sync   %.1f us" % t(b))
def c(): st["x"].copy_(x); st["y"].copy_(y); st["graph"].replay(); s.synchronize()
print("2 copy_ + replay + sync     %.1f us" % t(c))
def d(): be.copy_pair(st["x"], x, st["y"], y); st["graph"].replay(); s.synchronize(); st["host"].tolist()
print("... + tolist                %.1f us" % t(d))
print("_graph_key                  %.1f us" % t(lambda: net._graph_key(x, y)))
print("_prepare                    %.1f us" % t(lambda: net._prepare(x, y)))
print("set_mode_train              %.1f us" % t(lambda: net.set_mode_train(train=True)))
def e(): st["graph"].replay()
print("replay only (async, avg)    %.1f us" % t(e))
