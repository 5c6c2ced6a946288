"""Per-launch time of dun_fc_block_norm's statistics kernel (bn_reduce_finalize) at the wide configuration, for A/B
library builds (DUN_HIP_LIB).  GPU only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from bench import WORKLOADS, timed
from dun_amd import kernels
bench.import_torch()
din, W, L, B, N_train, lr, momentum, wd = WORKLOADS["c3_wide"]
net = bench.build_net(din, W, 2, N_train, lr, momentum, wd)
x, y = bench.shard_data(B, din, 0, torch.device("cuda", 0))
net.fit(x, y)
be = kernels.backend()
buf = net.model._get_buffers(B, 2, True)
rm, rv = net.model.running_mean.clone(), net.model.running_var.clone()
be.fc_block_linear(buf, 0)
full = timed(lambda: be.fc_block_norm(buf, 0, float(B)), 50)
apply_only = timed(lambda: be.fc_block_apply(buf, 0), 50)
print("%s: block_norm %.1f us, apply alone %.1f us -> statistics kernel %.1f us" % (
    os.environ.get("DUN_HIP_LIB", "default"), full * 1e3, apply_only * 1e3, (full - apply_only) * 1e3))
