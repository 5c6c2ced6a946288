"""Depth-uncertain ResNets with the reference's constructor surface (src/DUN/stochastic_img_resnets.py:134-359).

Scope: the convolutional trunk stays on MIOpen through torch -- blocks, per-stage
1x1 "adapt" stacks and the shared tail run once per exit with their own BatchNorm batch statistics, exactly as
the reference does (F7).  What changes is the end of every exit: the pooled features of all D exits are stacked
into one [D,B,K] tensor and the shared ``output_block`` is applied to all D*B rows as ONE batched contraction by
the HIP head kernels (forward and backward, dun_amd/csrc/head.hip); the per-depth softmax cross-entropy / ELBO on
the resulting ``act_vec`` is the fused kernel of dun_amd/probability.py.

Module names and registration order follow the reference so that ``state_dict()`` keys are interchangeable and
the same torch seed yields the same initial weights.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.modules.dropout import _DropoutNd

from . import kernels


__all__ = ['ResNet', 'ArchUncertResNet', 'resnet18', 'resnet34', 'resnet50', 'resnet101']


class MC_Dropout2d(_DropoutNd):
    """Channel dropout that stays active in eval mode (MC dropout)."""

    def forward(self, input):
        return F.dropout2d(input, self.p, True, self.inplace)


class AdaptiveConcatPool2d(nn.Module):
    def __init__(self, sz=None):
        super().__init__()
        sz = sz or (1, 1)
        self.ap = nn.AdaptiveAvgPool2d(sz)
        self.mp = nn.AdaptiveMaxPool2d(sz)

    def forward(self, x):
        return torch.cat([self.mp(x), self.ap(x)], 1)


def conv3x3(cin, cout, stride=1):
    return nn.Conv2d(cin, cout, kernel_size=3, stride=stride, padding=1, bias=False)


def conv1x1(cin, cout, stride=1):
    return nn.Conv2d(cin, cout, kernel_size=1, stride=stride, bias=False)


def _drop(p):
    return MC_Dropout2d(p=p, inplace=False) if p > 0 else nn.Identity()


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, p_drop=0):
        super(BasicBlock, self).__init__()
        self.p_drop = p_drop
        self.drop_layer = _drop(p_drop)
        self.conv1 = conv3x3(inplanes, planes, stride)  # this conv and `downsample` carry the stride
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = conv3x3(planes, planes)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        skip = x if self.downsample is None else self.downsample(x)
        out = self.drop_layer(self.relu(self.bn1(self.conv1(x))))
        out = self.bn2(self.conv2(out))
        out += skip
        return self.relu(out)


class Bottleneck(nn.Module):
    expansion = 4  # stride on the 3x3 conv ("ResNet v1.5")

    def __init__(self, inplanes, planes, stride=1, downsample=None, p_drop=0):
        super(Bottleneck, self).__init__()
        self.p_drop = p_drop
        self.drop_layer = _drop(p_drop)
        self.conv1 = conv1x1(inplanes, planes)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = conv3x3(planes, planes, stride)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = conv1x1(planes, planes * self.expansion)
        self.bn3 = nn.BatchNorm2d(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        skip = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.drop_layer(self.relu(self.bn2(self.conv2(out))))
        out = self.bn3(self.conv3(out))
        out += skip
        return self.relu(out)


class _SharedHead(torch.autograd.Function):
    """act[d,b,:] = W feat[d,b,:] + bias for all D*B rows at once (HIP head kernels, forward and backward)."""

    @staticmethod
    def forward(ctx, feat, weight, bias):
        D, B, K = feat.shape
        f2 = feat.detach().to(torch.float32).contiguous().view(D * B, K)
        act = kernels.backend().head_forward(f2, weight.detach().contiguous(), bias.detach().contiguous())
        ctx.save_for_backward(f2, weight)
        ctx.shape = (D, B, K)
        return act.view(D, B, -1)

    @staticmethod
    def backward(ctx, dact):
        f2, weight = ctx.saved_tensors
        D, B, K = ctx.shape
        d2 = dact.detach().to(torch.float32).contiguous().view(D * B, -1)
        dw, db, dH = kernels.backend().head_backward(f2, weight.detach().contiguous(), d2,
                                                     need_dH=ctx.needs_input_grad[0])
        return (None if dH is None else dH.view(D, B, K)), dw, db


class ResNet(nn.Module):
    def __init__(self, block, layers, num_classes=1000, zero_init_residual=True, initial_conv='1x7',
                 concat_pool=False, input_chanels=3, p_drop=0):
        super(ResNet, self).__init__()
        self.zero_init_residual = zero_init_residual
        self.num_classes = num_classes
        self._norm_layer = nn.BatchNorm2d
        self.inplanes = 64
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)

        if initial_conv == '1x7':
            stem = [nn.Conv2d(input_chanels, 64, kernel_size=7, stride=2, padding=3, bias=False),
                    nn.BatchNorm2d(64), self.relu, self.maxpool]
        elif initial_conv == '1x3':
            stem = [conv3x3(input_chanels, 64), nn.BatchNorm2d(64), self.relu]
        elif initial_conv == '3x3':
            stem = []
            for cin in (input_chanels, 64, 64):
                stem += [conv3x3(cin, 64), nn.BatchNorm2d(64), self.relu]
            stem.append(self.maxpool)
        else:
            raise ValueError("initial_conv must be '1x7', '1x3' or '3x3'")
        self.conv1 = nn.Sequential(*stem)

        self.layer_list = nn.ModuleList()
        for planes, n, stride in zip((64, 128, 256, 512), layers, (1, 2, 2, 2)):
            self.layer_list += self._make_layer(block, planes, n, stride=stride, p_drop=p_drop)
        self.n_layers = len(self.layer_list)

        feat = 512 * block.expansion
        if concat_pool:
            self.pool = AdaptiveConcatPool2d((1, 1))
            self.output_block = nn.Linear(2 * feat, self.num_classes)
        else:
            self.pool = nn.AdaptiveAvgPool2d((1, 1))
            self.output_block = nn.Linear(feat, self.num_classes)
        self._init_layers()

    def _init_layers(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        if self.zero_init_residual:  # residual branches start as identities (arXiv 1706.02677)
            for m in self.modules():
                if isinstance(m, Bottleneck):
                    nn.init.constant_(m.bn3.weight, 0)
                elif isinstance(m, BasicBlock):
                    nn.init.constant_(m.bn2.weight, 0)

    def _make_layer(self, block, planes, blocks, stride=1, p_drop=0):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(conv1x1(self.inplanes, planes * block.expansion, stride),
                                       nn.BatchNorm2d(planes * block.expansion))
        # (as in the reference, the first block of a stage gets no dropout)
        out = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        out += [block(self.inplanes, planes, p_drop=p_drop) for _ in range(1, blocks)]
        return out

    def _pooled(self, x):
        return torch.flatten(self.pool(x), 1)

    def _forward_impl(self, x):
        x = self.conv1(x)
        for layer in self.layer_list:
            x = layer(x)
        return self.output_block(self._pooled(x))

    def forward(self, x):
        return self._forward_impl(x)


class ArchUncertResNet(ResNet):
    def __init__(self, block, layers, start_depth, end_depth, num_classes=1000, zero_init_residual=True,
                 initial_conv='1x7', concat_pool=False, input_chanels=3, p_drop=0):
        super(ArchUncertResNet, self).__init__(block, layers, num_classes=num_classes,
                                               zero_init_residual=zero_init_residual, initial_conv=initial_conv,
                                               concat_pool=concat_pool, input_chanels=input_chanels, p_drop=p_drop)
        self.start_depth = start_depth
        self.end_depth = end_depth
        self.n_layers = self.end_depth - self.start_depth
        self.channel_list = [s for s, n in enumerate(layers) for _ in range(n)]  # stage index of every block

        e = block.expansion
        self.adapt0 = nn.Sequential(conv1x1(64 * e, 128 * e, stride=2), nn.BatchNorm2d(128 * e), self.relu)
        self.adapt1 = nn.Sequential(conv1x1(128 * e, 256 * e, stride=2), nn.BatchNorm2d(256 * e), self.relu)
        self.adapt2 = nn.Sequential(conv1x1(256 * e, 512 * e, stride=2), nn.BatchNorm2d(512 * e), self.relu)
        self.adapt3 = nn.Identity()
        # an exit in stage s is brought to the shape of the stage that holds block `end_depth` by chaining adapts
        ends = [sum(layers[:i + 1]) for i in range(4)]
        last = next((i for i, c in enumerate(ends) if self.end_depth <= c), 3)
        adapts = [self.adapt0, self.adapt1, self.adapt2]
        stacks = []
        for s in range(last):
            chain = adapts[s:last]
            stacks.append(chain[0] if len(chain) == 1 else nn.Sequential(*chain))
        stacks.append(self.adapt3)
        self.adapt_layers = nn.ModuleList(stacks)
        self._init_layers()

    def fwd_input_block(self, x):
        x = self.conv1(x)
        for layer in self.layer_list[:self.start_depth]:
            x = layer(x)
        return x

    def _exit_features(self, x, layer_idx):
        y = self.adapt_layers[self.channel_list[layer_idx]](x)
        for layer in self.layer_list[self.end_depth:]:
            y = layer(y)
        return self._pooled(y)

    def fwd_output_block(self, x):
        for layer in self.layer_list[self.end_depth:]:
            x = layer(x)
        return self.output_block(self._pooled(x))

    def _head(self, feats):
        """Shared Linear over the stacked [D,B,K] exit features: one batched HIP contraction."""
        return _SharedHead.apply(torch.stack(feats, dim=0), self.output_block.weight, self.output_block.bias)

    def fast_forward_impl(self, x, layer_probs, min_prob=1e-2, device_mask=False):
        """Early-exit inference: exits whose posterior mass is <= min_prob are not evaluated (their rows stay 0).
        One device->host copy of the D probabilities decides all exits (the reference syncs once per exit,
        src/DUN/stochastic_img_resnets.py:296).

        device_mask=True: NO host synchronisation at all -- every exit is evaluated and the rows of the low-mass
        exits are zeroed on the device by the same strict comparison.  Same act_vec (kept rows to fp32 rounding of a
        differently shaped head contraction, exact zeros for the others), nothing skipped: for callers that cannot synchronise (a captured serving
        step), where the decision which blocks to run cannot be taken on the host anyway."""
        if device_mask:
            act_vec = self._forward_impl(x)
            keep = (layer_probs.detach() > min_prob).to(act_vec.device).view(-1, 1, 1)
            return torch.where(keep, act_vec, torch.zeros_like(act_vec))
        keep = (layer_probs.detach() > min_prob).tolist()
        x = self.fwd_input_block(x)
        feats, kept = [], []
        for idx, layer_idx in enumerate(range(self.start_depth, self.end_depth)):
            x = self.layer_list[layer_idx](x)
            if keep[idx]:
                feats.append(self._exit_features(x, layer_idx))
                kept.append(idx)
            else:
                print('skipping layer %d' % layer_idx)
        act_vec = x.new_zeros(self.n_layers, x.shape[0], self.num_classes)
        if feats:
            act_vec[kept] = self._head(feats)
        return act_vec

    def _forward_impl(self, x):
        x = self.fwd_input_block(x)
        feats = []
        for layer_idx in range(self.start_depth, self.end_depth):
            x = self.layer_list[layer_idx](x)
            feats.append(self._exit_features(x, layer_idx))
        return self._head(feats)

    def forward(self, x, depth=None):
        return self._forward_impl(x)

    def get_w_prior_loglike(self, k=0):
        return self.adapt0[0].weight.data.new_zeros(self.n_layers)


def _resnet(block, layers, arch_uncert=True, **kwargs):
    return ArchUncertResNet(block, layers, **kwargs) if arch_uncert else ResNet(block, layers, **kwargs)


def resnet18(**kwargs):
    return _resnet(BasicBlock, [2, 2, 2, 2], **kwargs)


def resnet34(**kwargs):
    return _resnet(BasicBlock, [3, 4, 6, 3], **kwargs)


def resnet50(**kwargs):
    return _resnet(Bottleneck, [3, 4, 6, 3], **kwargs)


def resnet101(**kwargs):
    return _resnet(Bottleneck, [3, 4, 23, 3], **kwargs)
