"""Oracle: ResNet regression trunk + deconv head as a pure function of a state dict (TEST INFRASTRUCTURE).

Restates ``ResNet.forward`` (reference lib/models/backbones/Resnet.py:56-67: conv7x7 s2 - BN - ReLU - maxpool 3x3 s2 -
layer1..4 of Bottlenecks, stride on the 3x3 conv, 1x1 stride-s projection on the first block of a layer,
Resnet.py:40-54, 96-135) and the deconv head of ``RootNetwithRegInt`` (lib/models/full_net.py:194-216: three
ConvTranspose2d(4, stride 2, padding 1, bias=False) + BN + ReLU, then the 1x1 ``final_layer``), driven by the
reference's state-dict keys.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this.
"""
import torch.nn.functional as F

from .hrnet import _Ctx, _bn, _conv

LAYERS = {"resnet50": [3, 4, 6, 3], "resnet101": [3, 4, 23, 3]}   # Bottleneck variants (Resnet.py:9-13)


def _bottleneck(c, key, x, stride):
    """Resnet.py:113-135."""
    out = F.relu(_bn(c, key + ".bn1", _conv(c, key + ".conv1", x)))
    out = F.relu(_bn(c, key + ".bn2", _conv(c, key + ".conv2", out, stride=stride)))
    out = _bn(c, key + ".bn3", _conv(c, key + ".conv3", out))
    if c.has(key + ".downsample.0.weight"):
        x = _bn(c, key + ".downsample.1", _conv(c, key + ".downsample.0", x, stride=stride))
    return F.relu(out + x)


def resnet_forward(sd, x, prefix="", name="resnet50", training=False):
    """-> x_out [B, 2048, H/32, W/32] (Resnet.py:56-67)."""
    c = _Ctx(sd, prefix, training)
    h = F.relu(_bn(c, "bn1", _conv(c, "conv1", x, stride=2)))          # 7x7, padding 3 (= k // 2)
    h = F.max_pool2d(h, kernel_size=3, stride=2, padding=1)
    for li, n in enumerate(LAYERS[name]):
        for b in range(n):
            h = _bottleneck(c, f"layer{li + 1}.{b}", h, stride=2 if (b == 0 and li > 0) else 1)
    return h


def deconv_head_forward(sd, x_out, training=False):
    """full_net.py:293-296: heat-map logits = final_layer(deconv_layers(x_out)); xf = avgpool(x_out)."""
    c = _Ctx(sd, "", training)
    h = x_out
    for i in (0, 3, 6):
        h = F.conv_transpose2d(h, sd[f"deconv_layers.{i}.weight"], None, stride=2, padding=1)
        h = F.relu(_bn(c, f"deconv_layers.{i + 1}", h))
    heat = F.conv2d(h, sd["final_layer.weight"], sd["final_layer.bias"])
    xf = F.avg_pool2d(x_out, x_out.shape[-1], stride=1).flatten(1)
    return heat, xf
