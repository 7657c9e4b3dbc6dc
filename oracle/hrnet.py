"""Oracle: HRNet-W32 trunk + heads as a pure function of a state dict (TEST INFRASTRUCTURE).

Restates ``PoseHighResolutionNet.forward`` (reference lib/models/backbones/HRnet.py:499-570) and the
blocks it is made of, driven by the reference's state-dict key names instead of a module tree.
Topology constants come from lib/models/backbones/configs/hrnet_w32.yaml:54-93.
"""
import torch
import torch.nn.functional as F

# hrnet_w32.yaml:55-93 - (modules, channels per branch); every branch has 4 BASIC blocks, fuse = SUM
STAGES = {
    "stage2": (1, [32, 64]),
    "stage3": (4, [32, 64, 128]),
    "stage4": (3, [32, 64, 128, 256]),
}
BLOCKS_PER_BRANCH = 4
BN_EPS = 1e-5
BN_MOMENTUM = 0.1  # HRnet.py:18 (and nn.BatchNorm2d's default for the un-annotated ones)


class _Ctx:
    def __init__(self, sd, prefix, training):
        self.sd = sd
        self.prefix = prefix
        self.training = training

    def p(self, name):
        return self.sd[self.prefix + name]

    def has(self, name):
        return (self.prefix + name) in self.sd


def _bn(c, key, x):
    """nn.BatchNorm2d forward, train (batch stats + running update) or eval."""
    return F.batch_norm(x, c.p(key + ".running_mean"), c.p(key + ".running_var"),
                        c.p(key + ".weight"), c.p(key + ".bias"), c.training, BN_MOMENTUM, BN_EPS)


def _conv(c, key, x, stride=1):
    w = c.p(key + ".weight")
    b = c.p(key + ".bias") if c.has(key + ".bias") else None
    return F.conv2d(x, w, b, stride=stride, padding=w.shape[-1] // 2)


def _basic_block(c, key, x):
    """HRnet.py:41-57: conv3x3-BN-ReLU-conv3x3-BN-(+x)-ReLU (no downsample inside stages)."""
    out = F.relu(_bn(c, key + ".bn1", _conv(c, key + ".conv1", x)))
    out = _bn(c, key + ".bn2", _conv(c, key + ".conv2", out))
    return F.relu(out + x)


def _bottleneck(c, key, x):
    """HRnet.py:78-98: 1x1-BN-ReLU-3x3-BN-ReLU-1x1-BN-(+down(x))-ReLU."""
    out = F.relu(_bn(c, key + ".bn1", _conv(c, key + ".conv1", x)))
    out = F.relu(_bn(c, key + ".bn2", _conv(c, key + ".conv2", out)))
    out = _bn(c, key + ".bn3", _conv(c, key + ".conv3", out))
    if c.has(key + ".downsample.0.weight"):
        x = _bn(c, key + ".downsample.1", _conv(c, key + ".downsample.0", x))
    return F.relu(out + x)


def _hr_module(c, key, xs, multi_scale_output=True):
    """HighResolutionModule.forward, HRnet.py:247-265 with fuse layers built as in :187-242."""
    nb = len(xs)
    xs = list(xs)
    for b in range(nb):
        for k in range(BLOCKS_PER_BRANCH):
            xs[b] = _basic_block(c, f"{key}.branches.{b}.{k}", xs[b])
    outs = []
    for i in range(nb if multi_scale_output else 1):
        y = None
        for j in range(nb):
            fk = f"{key}.fuse_layers.{i}.{j}"
            if j == i:
                t = xs[j]
            elif j > i:
                # conv1x1 + BN + nearest upsample x2^(j-i)  (HRnet.py:197-208)
                t = _bn(c, fk + ".1", _conv(c, fk + ".0", xs[j]))
                t = F.interpolate(t, scale_factor=2 ** (j - i), mode="nearest")
            else:
                # (i-j) stride-2 3x3 convs; all but the last keep C_j and have ReLU (HRnet.py:211-239)
                t = xs[j]
                for k in range(i - j):
                    t = _bn(c, f"{fk}.{k}.1", _conv(c, f"{fk}.{k}.0", t, stride=2))
                    if k != i - j - 1:
                        t = F.relu(t)
            y = t if y is None else y + t
        outs.append(F.relu(y))
    return outs


def hrnet_w32_forward(sd, x, prefix="", generate_hm=True, generate_feat=True, training=False,
                      taps=None):
    """PoseHighResolutionNet.forward (HRnet.py:499-570).

    sd: mapping name -> tensor with the reference's key names under ``prefix``.
    taps: optional dict filled with intermediate activations (for per-stage parity checks).
    Returns heatmap, (heatmap, feat) or feat exactly like the reference.
    """
    c = _Ctx(sd, prefix, training)
    t = taps if taps is not None else {}
    x = F.relu(_bn(c, "bn1", _conv(c, "conv1", x, stride=2)))          # :500-502
    x = F.relu(_bn(c, "bn2", _conv(c, "conv2", x, stride=2)))          # :503-505
    t["stem"] = x
    for k in range(4):                                                # layer1, :506 / :291
        x = _bottleneck(c, f"layer1.{k}", x)
    t["layer1"] = x
    # transition1 (:508-513): branch0 conv3x3 256->32, branch1 conv3x3 s2 256->64
    ys = [F.relu(_bn(c, "transition1.0.1", _conv(c, "transition1.0.0", x))),
          F.relu(_bn(c, "transition1.1.0.1", _conv(c, "transition1.1.0.0", x, stride=2)))]
    ys = _hr_module(c, "stage2.0", ys)
    t["stage2"] = ys
    # transition2 (:516-521): new branch from the LAST branch output
    ys = ys + [F.relu(_bn(c, "transition2.2.0.1", _conv(c, "transition2.2.0.0", ys[-1], stride=2)))]
    for m in range(STAGES["stage3"][0]):
        ys = _hr_module(c, f"stage3.{m}", ys)
    t["stage3"] = ys
    ys = ys + [F.relu(_bn(c, "transition3.3.0.1", _conv(c, "transition3.3.0.0", ys[-1], stride=2)))]
    nmod = STAGES["stage4"][0]
    for m in range(nmod):
        # multi_scale_output = generate_feat on the last module only (:322-323, :477-482)
        ys = _hr_module(c, f"stage4.{m}", ys, multi_scale_output=(generate_feat or m != nmod - 1))
    t["stage4"] = ys
    heat = None
    if generate_hm:
        heat = _conv(c, "final_layer", ys[0])                         # :533
        if not generate_feat:
            return heat
    # classification head (:537-548)
    y = _bottleneck(c, "incre_modules.0.0", ys[0])
    for i in range(3):
        d = F.relu(_bn(c, f"downsamp_modules.{i}.1", _conv(c, f"downsamp_modules.{i}.0", y, stride=2)))
        y = _bottleneck(c, f"incre_modules.{i + 1}.0", ys[i + 1]) + d
    y = F.relu(_bn(c, "final_feat_layer.1", _conv(c, "final_feat_layer.0", y)))
    t["head_map"] = y
    feat = F.avg_pool2d(y, kernel_size=y.shape[2:]).view(y.shape[0], -1)
    if generate_hm:
        return heat, feat
    return feat
