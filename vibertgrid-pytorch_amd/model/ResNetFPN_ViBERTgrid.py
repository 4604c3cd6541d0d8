"""ResNet-FPN backbone with ViBERTgrid early fusion on MI355X kernels.

Parameter containers mirror the reference's module tree so `state_dict()` keys are identical
(reference model/ResNetFPN_ViBERTgrid.py: `BasicBlock` :106-184, `EarlyFusionLayer` :272-321,
`ResNetFPN_ViBERTgrid` :324-508, `ResNetFPN_ViBERTgrid_Pretrained` :511-648, factories :651-720).
The nn.Conv2d / nn.BatchNorm2d objects below only OWN parameters and buffers; the arithmetic is
libvbg: NHWC implicit-GEMM convolutions on fp32 MFMA with BatchNorm / residual / ReLU kernels, the
early-fusion and P_fuse 1x1 convolutions read their concatenated operands in place (no torch.cat,
no 67 MB/doc upsampled concat), FPN top-down adds are one fused upsample+add kernel.
"""
import warnings
from typing import List

import torch
import torch.nn as nn

from vbg import functions as Fn


def _cl(module: nn.Module):
    """put every 4-D parameter in channels_last memory = physically [Cout, kh, kw, Cin]"""
    for p in module.parameters():
        if p.dim() == 4:
            p.data = p.data.contiguous(memory_format=torch.channels_last)
    return module


class bn_tick_scope:
    """Inside the scope the `num_batches_tracked += 1` of every training-mode BatchNorm (torch.nn.BatchNorm2d.forward) is collected
    and applied by ONE multi-tensor launch on exit instead of one tiny launch per layer (40 at resnet34 + heads)."""
    _pending = None

    def __enter__(self):
        self._outer = bn_tick_scope._pending
        bn_tick_scope._pending = []
        return self

    def __exit__(self, *exc):
        ticks, bn_tick_scope._pending = bn_tick_scope._pending, self._outer
        if ticks:
            torch._foreach_add_(ticks, 1)
        return False


def conv_bn(x, conv: nn.Conv2d, bn: nn.Module, res=None, relu=True):
    if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
        if bn_tick_scope._pending is not None:
            bn_tick_scope._pending.append(bn.num_batches_tracked)
        else:
            bn.num_batches_tracked.add_(1)
    return Fn.ConvBnFn.apply(x, conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var, res, conv.stride[0],
                             conv.padding[0], relu, bn.training, bn.momentum, bn.eps, isinstance(bn, nn.SyncBatchNorm))


def conv(x, c: nn.Conv2d):
    return Fn.ConvFn.apply(x, c.weight, c.bias, c.stride[0], c.padding[0])


class BasicBlock(nn.Module):
    """two 3x3 conv+BN, identity or 1x1/s2 projection shortcut (reference key names)"""

    def __init__(self, in_channel: int, out_channel: int, downsample: bool = False) -> None:
        super().__init__()
        self.in_channel, self.out_channel = in_channel, out_channel
        if downsample:
            self.conv_1 = nn.Conv2d(in_channel, out_channel, 3, 2, 1, bias=False)
            self.conv_shortcut = nn.Sequential(nn.Conv2d(in_channel, out_channel, 1, 2, 0, bias=False), nn.BatchNorm2d(out_channel))
        else:
            self.conv_1 = nn.Conv2d(out_channel, out_channel, 3, 1, 1, bias=False)
            self.conv_shortcut = nn.Identity()
        self.bn_1 = nn.BatchNorm2d(out_channel)
        self.conv_2 = nn.Conv2d(out_channel, out_channel, 3, 1, 1, bias=False)
        self.bn_2 = nn.BatchNorm2d(out_channel)

    def forward(self, x):
        y = conv_bn(x, self.conv_1, self.bn_1, None, True)
        s = x if isinstance(self.conv_shortcut, nn.Identity) else conv_bn(x, self.conv_shortcut[0], self.conv_shortcut[1], None, False)
        return conv_bn(y, self.conv_2, self.bn_2, s, True)


class DBlock(BasicBlock):
    """ResNet-D flavour of the basic block (reference :187-269): the projection shortcut is
    AvgPool2d(2, 2) -> 1x1 conv (stride 1) -> BN, so `conv_shortcut` holds (pool, conv, bn) = keys .1 / .2"""

    def __init__(self, in_channel: int, out_channel: int, downsample: bool = False) -> None:
        super().__init__(in_channel, out_channel, downsample)
        if downsample:
            self.conv_shortcut = nn.Sequential(nn.AvgPool2d(kernel_size=2, stride=2),
                                               nn.Conv2d(in_channel, out_channel, 1, 1, 0, bias=False), nn.BatchNorm2d(out_channel))

    def forward(self, x):
        y = conv_bn(x, self.conv_1, self.bn_1, None, True)
        if isinstance(self.conv_shortcut, nn.Identity):
            s = x
        else:
            s = conv_bn(Fn.AvgPool2Fn.apply(x), self.conv_shortcut[1], self.conv_shortcut[2], None, False)
        return conv_bn(y, self.conv_2, self.bn_2, s, True)


class EarlyFusionLayer(nn.Module):
    def __init__(self, block, in_channel: int, out_channel: int, block_num: int, grid_channel: int, downsample=True) -> None:
        super().__init__()
        self.block_1 = block(in_channel, out_channel, downsample=downsample)
        self.early_fusion = nn.Conv2d(out_channel + grid_channel, out_channel, kernel_size=1)       # bias=True (:305-309)
        self.layers = nn.Sequential(*[block(in_channel, out_channel, downsample=False) for _ in range(block_num - 1)])

    def fuse(self, x, grid):
        """early fusion of the (already computed) block_1 output with the grid, then the rest of the layer"""
        w = self.early_fusion.weight
        x = Fn.SegLinearFn.apply(w, self.early_fusion.bias, (0, 0), (0, 0), x, grid)
        return self.layers(x)

    def forward(self, x, grid):
        return self.fuse(self.block_1(x), grid)


class _FPN(nn.Module):
    """top-down pathway + P_fuse shared by both backbone flavours (:393-464 / :539-610)"""

    def _build_fpn(self, pyramid_channel: int, fuse_channel: int):
        self.conv_6_x = nn.Conv2d(512, pyramid_channel, 1, 1, 0, bias=False)
        for j, c in ((1, 256), (2, 128), (3, 64)):
            setattr(self, f"skip_{j}", nn.Conv2d(c, pyramid_channel, 1, 1, 0, bias=False))
            setattr(self, f"upsample_{j}", nn.Upsample(scale_factor=2, mode="nearest"))
            setattr(self, f"merge_{j}", nn.Conv2d(pyramid_channel, pyramid_channel, 3, 1, 1, bias=False))
        self.upsample_4 = nn.Upsample(scale_factor=2, mode="nearest")
        self.fuse_up_1 = nn.Upsample(scale_factor=8, mode="nearest")
        self.fuse_up_2 = nn.Upsample(scale_factor=4, mode="nearest")
        self.fuse_up_3 = nn.Upsample(scale_factor=2, mode="nearest")
        self.fuse = nn.Conv2d(4 * pyramid_channel, fuse_channel, 1, 1, 0, bias=False)

    def _fpn(self, x_1, x_2, x_3, c5):
        x_4 = conv(c5, self.conv_6_x)
        x_5 = conv(Fn.UpAddFn.apply(x_4, conv(x_3, self.skip_1)), self.merge_1)
        x_6 = conv(Fn.UpAddFn.apply(x_5, conv(x_2, self.skip_2)), self.merge_2)
        x_7 = conv(Fn.UpAddFn.apply(x_6, conv(x_1, self.skip_3)), self.merge_3)
        w = self.fuse.weight
        hw = (x_7.shape[1], x_7.shape[2])
        return Fn.SegLinearFn.apply(w, None, (3, 2, 1, 0), hw, x_4, x_5, x_6, x_7)


class ResNetFPN_ViBERTgrid(_FPN):
    def __init__(self, block, size_list: List, grid_channel: int, pyramid_channel: int = 256, fuse_channel: int = 256) -> None:
        super().__init__()
        self.conv_1 = nn.Sequential(nn.Conv2d(3, 64, 7, 2, 3, bias=False), nn.BatchNorm2d(64), nn.ReLU(inplace=True))
        self.pool_1 = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.conv_2_x = self._make_layer(block, 64, 64, size_list[0], False)
        self.conv_3_x = EarlyFusionLayer(block, 64, 128, size_list[1], grid_channel, True)
        self.conv_4_x = self._make_layer(block, 128, 256, size_list[2], True)
        self.conv_5_x = self._make_layer(block, 256, 512, size_list[3], True)
        self._build_fpn(pyramid_channel, fuse_channel)
        _cl(self)

    @staticmethod
    def _make_layer(block, in_channel, out_channel, block_num, downsample=True):
        return nn.Sequential(*[block(in_channel if i == 0 else out_channel, out_channel, downsample=(downsample and i == 0))
                               for i in range(block_num)])

    def stage1(self, input):
        """everything in front of the early fusion (independent of the BERTgrid): -> (C2, first block of conv_3_x)"""
        x_1 = conv_bn(input, self.conv_1[0], self.conv_1[1], None, True)
        x_1 = self.conv_2_x(Fn.MaxPoolFn.apply(x_1))
        return x_1, self.conv_3_x.block_1(x_1)

    def stage2(self, pre, grid):
        x_1, x_2 = pre
        x_2 = self.conv_3_x.fuse(x_2, grid)
        x_3 = self.conv_4_x(x_2)
        return self._fpn(x_1, x_2, x_3, self.conv_5_x(x_3))

    def forward(self, input, grid):
        """input NHWC [B,H,W,3]; grid NHWC [B,H/8,W/8,768] -> P_fuse NHWC [B,H/4,W/4,256]"""
        return self.stage2(self.stage1(input), grid)


class _TvBlock(nn.Module):
    """torchvision BasicBlock parameter layout (conv1, bn1, conv2, bn2, downsample.{0,1})"""

    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        y = conv_bn(x, self.conv1, self.bn1, None, True)
        s = x if self.downsample is None else conv_bn(x, self.downsample[0], self.downsample[1], None, False)
        return conv_bn(y, self.conv2, self.bn2, s, True)


class _TvResNet(nn.Module):
    """parameter tree of torchvision.models.resnet18/34 (incl. the unused `fc`, so checkpoints interchange)"""

    def __init__(self, sizes):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        cin = 64
        for li, (c, n) in enumerate(zip((64, 128, 256, 512), sizes), 1):
            blocks = []
            for i in range(n):
                blocks.append(_TvBlock(cin, c, 2 if (i == 0 and li > 1) else 1))
                cin = c
            setattr(self, f"layer{li}", nn.Sequential(*blocks))
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512, 1000)


def _tv_resnet(kind: str, sizes):
    net = _TvResNet(sizes)
    try:
        import torchvision
        src = getattr(torchvision.models, kind)(pretrained=True)
        net.load_state_dict(src.state_dict())
    except Exception as e:       # torchvision / weights unavailable offline
        warnings.warn(f"ImageNet weights for {kind} unavailable ({type(e).__name__}); backbone left randomly initialised")
    return net


class ResNetFPN_ViBERTgrid_Pretrained(_FPN):
    def __init__(self, resnet_type: str, grid_channel: int, pyramid_channel: int = 256, fuse_channel: int = 256) -> None:
        super().__init__()
        if resnet_type == "resnet18":
            self.resnet = _tv_resnet("resnet18", [2, 2, 2, 2])
        elif resnet_type == "resnet34":
            self.resnet = _tv_resnet("resnet34", [3, 4, 6, 3])
        else:
            raise ValueError("invalid value of resnet_type")
        self.norm_fuse_channel = 128
        self.early_fusion = nn.Conv2d(grid_channel + 128, 128, kernel_size=1, stride=1, bias=False)     # no bias (:529-535)
        self.num_block_ly2 = len(self.resnet.layer2)
        self._build_fpn(pyramid_channel, fuse_channel)
        _cl(self)

    def stage1(self, input):
        """everything in front of the early fusion (independent of the BERTgrid): -> (C2, layer2[0] output)"""
        r = self.resnet
        x_1 = r.layer1(Fn.MaxPoolFn.apply(conv_bn(input, r.conv1, r.bn1, None, True)))
        return x_1, r.layer2[0](x_1)

    def stage2(self, pre, BERTgrid):
        r = self.resnet
        x_1, x_2 = pre
        w = self.early_fusion.weight
        x_2 = Fn.SegLinearFn.apply(w, None, (0, 0), (0, 0), x_2, BERTgrid)
        for i in range(1, self.num_block_ly2):
            x_2 = r.layer2[i](x_2)
        x_3 = r.layer3(x_2)
        return self._fpn(x_1, x_2, x_3, r.layer4(x_3))

    def forward(self, input, BERTgrid):
        return self.stage2(self.stage1(input), BERTgrid)


def resnet_18_fpn(grid_channel: int, pretrained: bool = False) -> nn.Module:
    if pretrained:
        return ResNetFPN_ViBERTgrid_Pretrained("resnet18", grid_channel)
    return ResNetFPN_ViBERTgrid(BasicBlock, [2, 2, 2, 2], grid_channel)


def resnet_34_fpn(grid_channel: int, pretrained: bool = False) -> nn.Module:
    if pretrained:
        return ResNetFPN_ViBERTgrid_Pretrained("resnet34", grid_channel)
    return ResNetFPN_ViBERTgrid(BasicBlock, [3, 4, 6, 3], grid_channel)


def resnet_18_D_fpn(grid_channel: int) -> nn.Module:
    return ResNetFPN_ViBERTgrid(DBlock, [2, 2, 2, 2], grid_channel)


def resnet_34_D_fpn(grid_channel: int) -> nn.Module:
    return ResNetFPN_ViBERTgrid(DBlock, [3, 4, 6, 3], grid_channel)
