import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_net, shard_data
for (L, B) in ((50, 8192), (8, 65536), (50, 65536)):
    net = build_net(64, 2048, L, 1 << 20, 1e-2, 0.5, 0.0)
    x, y = shard_data(B, 64, 0, torch.device("cuda"))
    for s in range(3):
        r = net.fit(x, y)
        buf = net.model._get_buffers(B, L, True)
        print(L, B, s, r, "H finite", bool(torch.isfinite(buf.H[-1]).all()), "act absmax", float(buf.act.abs().max()),
              "gradmax", float(net.model.flat_grads.abs().max()), "pmax", float(net.model.flat_params.abs().max()), flush=True)
    del net, buf
    torch.cuda.empty_cache()
