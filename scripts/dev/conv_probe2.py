"""channels_last: which kernels answer conv -> bn -> relu -> conv (all 64 ch, 3x3, batch 256, 28x28) against conv alone"""
import sys, time
import torch
import torch.nn as nn
from torch.profiler import profile, ProfilerActivity
torch.backends.cudnn.benchmark = False
mode = sys.argv[1] if len(sys.argv) > 1 else "chain"
C = 64
if mode == "conv":
    net = nn.Sequential(nn.Conv2d(C, C, 3, padding=1, bias=False))
elif mode == "convconv":
    net = nn.Sequential(nn.Conv2d(C, C, 3, padding=1, bias=False), nn.Conv2d(C, C, 3, padding=1, bias=False))
elif mode == "convrelu":
    net = nn.Sequential(nn.Conv2d(C, C, 3, padding=1, bias=False), nn.ReLU(inplace=True), nn.Conv2d(C, C, 3, padding=1, bias=False))
else:
    net = nn.Sequential(nn.Conv2d(C, C, 3, padding=1, bias=False), nn.BatchNorm2d(C), nn.ReLU(inplace=True),
                        nn.Conv2d(C, C, 3, padding=1, bias=False))
net = net.cuda().to(memory_format=torch.channels_last)
x = torch.randn(256, C, 28, 28, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
def step():
    y = net(x)
    y.backward(torch.ones_like(y))
t0 = time.time(); step(); torch.cuda.synchronize()
print(mode, "first step %.1f s" % (time.time() - t0), flush=True)
step(); torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    step(); torch.cuda.synchronize()
for e in sorted(prof.key_averages(), key=lambda e: -e.device_time_total)[:7]:
    if e.device_time_total > 0:
        print("  %-90s n=%d %.3f ms" % (e.key[:90], e.count, e.device_time_total / e.count / 1e3))
