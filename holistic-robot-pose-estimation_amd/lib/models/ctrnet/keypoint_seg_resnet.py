"""``KeyPointSegNet`` (reference lib/models/ctrnet/keypoint_seg_resnet.py:103-149): torchvision's ``deeplabv3_resnet50`` with a
one-class head, executed as a static plan of HIP launches.  Same class name and the same state-dict keys as the reference
(``backbone.0.*`` = the dilated ResNet-50, ``classifer.0.*`` = torchvision's DeepLabHead - the reference's spelling -,
``read_out.kps_score_lowres.*``), so the authors' ``models/panda_segmentation/*.pth`` files load unchanged.

Only the SEGMENTATION branch is executed: the mask network's one caller, ``inference_batch_images_onlyseg`` (CtRNet.py:102-111),
discards the key-points; the key-point branch exists for CtRNet's BPnP pose solver (OpenCV on the host, SURVEY 2 out of scope).
``forward`` therefore returns ``(None, segout)``.

Architecture restated from torchvision 0.14 (models/segmentation/deeplabv3.py, models/resnet.py): ResNet-50 v1.5 with
``replace_stride_with_dilation=[False, True, True]`` (layer3 / layer4 keep 1/8 resolution with dilation 2 / 4), ASPP with rates
12 / 24 / 36 + image pooling, 3x3 conv, 1x1 conv.  Inference only (the trainer never trains this network:
scripts/train_sim2real.py:412 detaches its output).  PARITY UNPINNED: torchvision and the checkpoints are absent from the
reference tree and the image; tests compare with oracle/segnet.py, this repository's torch restatement of the same published
architecture.
"""
import torch
import torch.nn as nn

from hrpe_amd.plan import Term
from hrpe_amd.runtime import PlannedModule
from ..backbones.HRnet import BatchNorm2d, Conv2d, _Downsample, conv_bn
from ..backbones.Resnet import _StemConv

ASPP_RATES = (12, 24, 36)


class _Bottleneck(PlannedModule):
    """torchvision resnet.Bottleneck: 1x1 - 3x3 (stride, dilation) - 1x1 (x 4) + projection shortcut."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None):
        super().__init__()
        self.conv1 = Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = BatchNorm2d(planes)
        self.conv2 = Conv2d(planes, planes, 3, stride=stride, bias=False, dilation=dilation)
        self.bn2 = BatchNorm2d(planes)
        self.conv3 = Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = BatchNorm2d(planes * 4)
        self.downsample = downsample

    def emit(self, pb, x):
        h = pb.act([conv_bn(pb, x, self.conv1, self.bn1)], relu=True)
        h = pb.act([conv_bn(pb, h, self.conv2, self.bn2)], relu=True)
        skip = Term(x) if self.downsample is None else conv_bn(pb, x, self.downsample[0], self.downsample[1])
        return pb.act([conv_bn(pb, h, self.conv3, self.bn3), skip], relu=True)


class _DilatedResNet50(PlannedModule):
    """``deeplabv3_resnet50().backbone`` (an IntermediateLayerGetter over resnet50; its state-dict keys are the ResNet's)."""

    def __init__(self):
        super().__init__()
        self.inplanes, self.dilation = 64, 1
        self.conv1 = _StemConv()
        self.bn1 = BatchNorm2d(64)
        self.layer1 = self._make_layer(64, 3)
        self.layer2 = self._make_layer(128, 4, stride=2)
        self.layer3 = self._make_layer(256, 6, stride=2, dilate=True)
        self.layer4 = self._make_layer(512, 3, stride=2, dilate=True)

    def _make_layer(self, planes, blocks, stride=1, dilate=False):
        prev = self.dilation
        if dilate:                      # torchvision resnet._make_layer: the stride turns into dilation
            self.dilation *= stride
            stride = 1
        ds = None
        if stride != 1 or self.inplanes != planes * 4:
            ds = _Downsample(self.inplanes, planes * 4, stride)
        mods = [_Bottleneck(self.inplanes, planes, stride, prev, ds)]
        self.inplanes = planes * 4
        mods += [_Bottleneck(self.inplanes, planes, dilation=self.dilation) for _ in range(1, blocks)]
        return nn.Sequential(*mods)

    def emit(self, pb, xs):
        """xs: the normalised image in the stem's space-to-depth layout -> layer4 features [N, H/8, W/8, 2048]"""
        y = pb.stem7x7_s2d(xs, self.conv1.weight)
        h = pb.act([Term(y, self.bn1)], relu=True)
        h = pb.maxpool3x3s2(h)
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            for blk in layer:
                h = blk.emit(pb, h)
        return h


class _ASPP(PlannedModule):
    """torchvision deeplabv3.ASPP(2048, [12, 24, 36]): keys convs.{0..3}.{0,1}, convs.4.{1,2}, project.{0,1}."""

    def __init__(self, cin=2048, cout=256):
        super().__init__()
        convs = [nn.Sequential(Conv2d(cin, cout, 1, bias=False), BatchNorm2d(cout))]
        convs += [nn.Sequential(Conv2d(cin, cout, 3, bias=False, dilation=r), BatchNorm2d(cout)) for r in ASPP_RATES]
        convs.append(nn.Sequential(nn.Identity(), Conv2d(cin, cout, 1, bias=False), BatchNorm2d(cout)))     # AdaptiveAvgPool2d(1), conv, BN
        self.convs = nn.ModuleList(convs)
        self.project = nn.Sequential(Conv2d(5 * cout, cout, 1, bias=False), BatchNorm2d(cout))              # (+ ReLU, Dropout(0.5): eval)
        self.cout = cout

    def emit(self, pb, x):
        co = self.cout
        cat = pb.plan.new(x.N, x.H, x.W, 5 * co, x.dtype)          # torch.cat(res, dim=1): every branch writes its channel slice
        for i in range(4):
            conv, bn = self.convs[i][0], self.convs[i][1]
            sl = pb.channel_slice(cat, i * co, co)
            y = pb.conv(x, conv.weight, None, dilation=conv.dilation, out=sl)
            pb.act([Term(y, bn)], relu=True, out=sl)               # folded into the conv's epilogue, or (rates 12 / 24 / 36) in place
        # image pooling: global average -> 1x1 conv -> BN -> ReLU -> "bilinear" up-sampling of a 1 x 1 map = the vector everywhere
        pooled = pb.avgpool(x)
        pconv, pbn = self.convs[4][1], self.convs[4][2]
        v = pb.act([Term(pconv.emit(pb, pooled), pbn)], relu=True)
        pb.broadcast_hw(v, pb.channel_slice(cat, 4 * co, co))
        return pb.act([conv_bn(pb, cat, self.project[0], self.project[1])], relu=True)


class KeypointUpSample(PlannedModule):
    """Parameter holder of the key-point read-out (keypoint_seg_resnet.py:11-33: ConvTranspose2d(2048, k, 4, stride 2, padding 1)
    with bias).  Not executed (module docstring); present so that the reference's checkpoints load with strict keys."""

    def __init__(self, in_channels, num_keypoints):
        super().__init__()
        self.kps_score_lowres = nn.Module()
        self.kps_score_lowres.weight = nn.Parameter(torch.zeros(in_channels, num_keypoints, 4, 4))
        self.kps_score_lowres.bias = nn.Parameter(torch.zeros(num_keypoints))
        nn.init.kaiming_normal_(self.kps_score_lowres.weight, mode="fan_out", nonlinearity="relu")


class KeyPointSegNet(PlannedModule):
    def __init__(self, args, lim=[-1., 1., -1., 1.], use_gpu=True):
        super().__init__()
        self.args, self.lim = args, lim
        self.device = "cuda" if use_gpu else "cpu"
        self.backbone = nn.Sequential(_DilatedResNet50())
        self.read_out = KeypointUpSample(2048, args.n_kp)
        # torchvision DeepLabHead = Sequential(ASPP, Conv2d 3x3, BatchNorm2d, ReLU, Conv2d 1x1 -> one class (keypoint_seg_resnet.py:119))
        self.classifer = nn.Sequential(nn.Sequential(_ASPP(), Conv2d(256, 256, 3, bias=False), BatchNorm2d(256), nn.Identity(),
                                                     Conv2d(256, 1, 1, bias=True)))

    def emit_logits(self, pb, xs):
        """xs: normalised image, space-to-depth -> logits at 1/8 resolution [N, H/8, W/8, 1]"""
        feat = self.backbone[0].emit(pb, xs)
        head = self.classifer[0]
        y = head[0].emit(pb, feat)
        y = pb.act([conv_bn(pb, y, head[1], head[2])], relu=True)
        return head[4].emit(pb, y)

    def _build(self, pb, img):
        if pb.plan.need_grad or pb.plan.training:
            raise NotImplementedError("KeyPointSegNet is an inference network here (eval(), no gradients): the reference never trains it")
        N, Cc, H, W = img.shape
        t = pb.image_input_s2d("img", N, Cc, H, W, u8=False)
        logits = self.emit_logits(pb, t)
        holder = pb.bilinear_nchw_output(logits, H, W)              # F.interpolate(x, size=input_shape, mode='bilinear', align_corners=False)
        holder["handle"] = logits
        return ["img"], [("nchw", holder, None)], {"img": t}

    def forward(self, img):
        """-> (None, segout [B, 1, H, W]); see the module docstring for the key-points."""
        return None, self._run(img)[0]
