"""ResNet regression trunk of the shipped ``full.yaml`` (reference lib/models/backbones/Resnet.py:5-135), same
class / factory names and state-dict keys, executed as a static plan of HIP launches.

The 7x7 stride-2 stem runs as a 4x4 stride-1 convolution over the 2x2 space-to-depth image (one 16-channel chunk of
the MFMA conv kernel instead of 49 taps over 3 channels, see PlanBuilder.stem7x7_s2d); max-pool, the Bottleneck
stacks (stride on the 3x3 conv, 1x1 stride-2 projection) and train-mode BatchNorm use the same kernels as HRNet.
"""
import torch
import torch.nn as nn

from hrpe_amd.plan import Term
from hrpe_amd.runtime import PlannedModule
from .HRnet import BN_MOMENTUM, BasicBlock, BatchNorm2d, Bottleneck, _Downsample, _emit_seq, conv_bn

_SPEC = {"resnet18": (BasicBlock, [2, 2, 2, 2]), "resnet34": (BasicBlock, [3, 4, 6, 3]),
         "resnet50": (Bottleneck, [3, 4, 6, 3]), "resnet101": (Bottleneck, [3, 4, 23, 3]),
         "resnet152": (Bottleneck, [3, 8, 36, 3])}


class _StemConv(PlannedModule):
    """Parameter holder of ``conv1`` = nn.Conv2d(3, 64, 7, stride 2, padding 3, bias=False) (Resnet.py:21)."""

    def __init__(self):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(64, 3, 7, 7))
        nn.init.normal_(self.weight, mean=0, std=0.001)   # Resnet.py:31-34
        self.bias = None


class ResNet(PlannedModule):
    def __init__(self, resnet_type):
        super().__init__()
        block, layers = _SPEC[resnet_type]
        self.block, self.name, self.inplanes = block, resnet_type, 64
        self.conv1 = _StemConv()
        self.bn1 = BatchNorm2d(64, momentum=BN_MOMENTUM)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        for m in self.modules():   # Resnet.py:31-37
            if hasattr(m, "kernel_size") and hasattr(m, "weight"):
                nn.init.normal_(m.weight, mean=0, std=0.001)

    def _make_layer(self, block, planes, blocks, stride=1):
        ds = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            ds = _Downsample(self.inplanes, planes * block.expansion, stride)
        mods = [block(self.inplanes, planes, stride, ds)]
        self.inplanes = planes * block.expansion
        mods += [block(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*mods)

    # -- plan description: `xs` is the space-to-depth image (PlanBuilder.image_input_s2d) ----------------------
    def emit(self, pb, xs):
        y = pb.stem7x7_s2d(xs, self.conv1.weight, want_stats=pb.plan.training)
        h = pb.act([Term(y, self.bn1)], relu=True)
        h = pb.maxpool3x3s2(h)
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            h = _emit_seq(pb, layer, h)
        return h

    def _build(self, pb, x):
        N, Cc, H, W = x.shape
        t = pb.image_input_s2d("x", N, Cc, H, W, u8=x.dtype == torch.uint8)
        y = self.emit(pb, t)
        holder = pb.nchw_output(y)
        holder["handle"] = y
        return ["x"], [("nchw", holder, None)], {"x": t}

    def forward(self, x):
        return self._run(x)[0]

    def init_weights(self, backbone_name):
        """The reference copies torchvision's ImageNet weights (Resnet.py:69-92); there is no network here, so the
        random initialisation above stays (state dicts of the reference load unchanged: identical keys)."""
        return None


def get_resnet(backbone_name, pretrain=True):
    """reference Resnet.py:182-193."""
    model = ResNet("resnet50" if backbone_name == "resnet" else backbone_name)
    if pretrain:
        model.init_weights(backbone_name)
    return model
