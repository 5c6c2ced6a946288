"""Which MIOpen kernels answer one trunk convolution (fp32), NCHW against channels_last, forward + backward; ms per call.
usage: python scripts/dev/conv_probe.py [cl] [Cin] [HW] [k] [Cout] [stride]"""
import sys, time
import torch
cl = "cl" in sys.argv
nums = [int(a) for a in sys.argv[1:] if a.isdigit()]
C = nums[0] if len(nums) > 0 else 64
HW = nums[1] if len(nums) > 1 else 32
k = nums[2] if len(nums) > 2 else 3
Co = nums[3] if len(nums) > 3 else C
st = nums[4] if len(nums) > 4 else 1
torch.backends.cudnn.benchmark = False
conv = torch.nn.Conv2d(C, Co, k, stride=st, padding=k // 2, bias=False).cuda()
x = torch.randn(4096, C, HW, HW, device="cuda", requires_grad=True)
if cl:
    conv = conv.to(memory_format=torch.channels_last)
    x = x.detach().contiguous(memory_format=torch.channels_last).requires_grad_(True)
def step():
    y = conv(x)
    y.backward(y)
t0 = time.time()
step(); torch.cuda.synchronize()
print("first step %.1f s" % (time.time() - t0), flush=True)
step(); torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(3):
        step()
    torch.cuda.synchronize()
rows = sorted(prof.key_averages(), key=lambda e: -e.device_time_total)[:5]
for e in rows:
    print("%-100s n=%d %.3f ms" % (e.key[:100], e.count, e.device_time_total / e.count / 1e3))
print("total per step %.2f ms" % (sum(e.device_time_total for e in prof.key_averages()) / 3e3), flush=True)
