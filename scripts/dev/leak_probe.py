"""Who keeps a dropped DUN_VI alive?  (GPU box)  Prints the referrer chain of the model after `del net; gc.collect()`."""
import gc, os, sys, weakref
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["DUN_NO_GRAPH"] = "1"
import torch
import bench
net = bench.build_net(16, 128, 4, 4096, 1e-3, 0.5, 0.0)
x, y = bench.shard_data(4096, 16, 0, torch.device("cuda"))
net.fit(x, y); net.fit(x, y)
wm, wn = weakref.ref(net.model), weakref.ref(net)
buf = list(net.model._fc_cache.values())[0]
wb = weakref.ref(buf)
del buf, net
gc.collect()
print("net alive:", wn() is not None, " model alive:", wm() is not None, " buf alive:", wb() is not None)
def show(obj, depth=0, seen=None):
    seen = seen if seen is not None else set()
    if depth > 4 or id(obj) in seen:
        return
    seen.add(id(obj))
    for r in gc.get_referrers(obj):
        if r is seen or isinstance(r, type(sys._getframe())) or id(r) in seen:
            continue
        desc = type(r).__name__
        if isinstance(r, dict):
            desc += " keys=" + ",".join(str(k) for k in list(r)[:8])
        elif isinstance(r, (list, tuple)):
            desc += " len=%d" % len(r)
        print("  " * depth + "<- " + desc[:160])
        show(r, depth + 1, seen)
for name, w in (("model", wm), ("buf", wb)):
    o = w()
    if o is not None:
        print("referrers of", name)
        show(o)
        del o
