"""Strides that reach the trunk's convolutions when bench.py's --channels-last path is taken (first six Conv2d calls)."""
import sys
import torch
sys.path.insert(0, ".")
import bench
torch.backends.cudnn.benchmark = False
net = bench.build_image_net(1, 60000)
torch.backends.cudnn.benchmark = False
net.model.to(memory_format=torch.channels_last)
x = torch.randn(256, 1, 28, 28).cuda().contiguous(memory_format=torch.channels_last)
y = torch.randint(0, 10, (256,)).cuda()
seen = []
def hook(m, inp):
    if len(seen) < 8:
        t = inp[0]
        seen.append(1)
        print("conv in %s strides %s CL=%s | weight %s strides %s CL=%s" % (tuple(t.shape), t.stride(), t.is_contiguous(memory_format=torch.channels_last),
              tuple(m.weight.shape), m.weight.stride(), m.weight.is_contiguous(memory_format=torch.channels_last)), flush=True)
for m in net.model.modules():
    if isinstance(m, torch.nn.Conv2d):
        m.register_forward_pre_hook(hook)
net.fit(x, y)
torch.cuda.synchronize()
print("benchmark flag", torch.backends.cudnn.benchmark, "deterministic", torch.backends.cudnn.deterministic)
