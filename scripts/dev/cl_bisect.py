"""One process, channels_last: a standalone 64->64 3x3 convolution step, then the same shape as the first blocks of the
image DUN see it -- which kernels answer each?"""
import sys
import torch
import torch.nn as nn
from torch.profiler import profile, ProfilerActivity
sys.path.insert(0, ".")
torch.backends.cudnn.benchmark = False

def kernels(fn, tag):
    fn(); torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        fn(); torch.cuda.synchronize()
    names = sorted(((e.device_time_total, e.key) for e in prof.key_averages() if e.device_time_total > 0), reverse=True)[:5]
    print("==", tag, flush=True)
    for t, k in names:
        print("   %8.3f ms %s" % (t / 1e3, k[:80]), flush=True)

order = sys.argv[1] if len(sys.argv) > 1 else "conv_first"
conv = nn.Conv2d(64, 64, 3, padding=1, bias=False).cuda().to(memory_format=torch.channels_last)
xc = torch.randn(256, 64, 28, 28, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
def conv_step():
    y = conv(xc); y.backward(torch.ones_like(y))
import bench
def model_step_factory():
    net = bench.build_image_net(1, 60000)
    torch.backends.cudnn.benchmark = False
    net.model.to(memory_format=torch.channels_last)
    x = torch.randn(256, 1, 28, 28).cuda().contiguous(memory_format=torch.channels_last)
    y = torch.randint(0, 10, (256,)).cuda()
    return lambda: net.fit(x, y)
if order == "conv_first":
    kernels(conv_step, "standalone conv (before the model exists)")
    ms = model_step_factory()
    kernels(ms, "model fit")
    kernels(conv_step, "standalone conv (after the model ran)")
else:
    ms = model_step_factory()
    kernels(ms, "model fit")
    kernels(conv_step, "standalone conv (after the model ran)")
