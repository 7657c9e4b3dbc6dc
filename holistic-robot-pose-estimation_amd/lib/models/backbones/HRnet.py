"""HRNet backbone on the hrpe_amd plan runtime (drop-in for reference lib/models/backbones/HRnet.py).

Same public names, constructor arguments, return values and state-dict keys as the reference:
``BasicBlock`` (HRnet.py:28-57), ``Bottleneck`` (:60-98), ``HighResolutionModule`` (:101-265),
``PoseHighResolutionNet`` (:274-603), ``load_hrnet_cfg`` (:606-610), ``get_hrnet`` (:613-623).
Nothing here computes with torch ops: every module describes itself to a PlanBuilder (``emit``) and the
plan executes hand-written HIP kernels (conv on MFMA, one-pass BN/add/upsample/ReLU fusion).
"""
import logging
import math
import os

import torch
import torch.nn as nn

from hrpe_amd.plan import Term
from hrpe_amd.runtime import PlannedModule, SingleTensorModule
from .configs import HRNET_CONFIGS, AttrDict

BN_MOMENTUM = 0.1  # reference HRnet.py:18
HEAD_FP32 = os.environ.get("HRP_HEAD_FP32", "0") not in ("0", "")
# measurement switch (DESIGN 4, bf16 key-point 0): from which stage on a FEATURE-ONLY trunk (the DepthNet) computes in fp32 - "" (bf16
# throughout), "4" or "3": the branch tensors entering that stage are cast and everything behind runs on the fp32 kernels; "1": the
# whole trunk from its input image on (full_net.py creates that input as an fp32 tensor)
HEAD_INCRE_LANES = os.environ.get("HRP_HEAD_INCRE_LANES", "1") not in ("0", "")      # emit_heads: the incre modules in virtual lanes
TRUNK_FP32_FROM = ""      # (a module constant: measurement tools set it before the plan is built)
logger = logging.getLogger(__name__)


# ---- parameter containers ----------------------------------------------------------------------------
class Conv2d(SingleTensorModule):
    """Holds [Cout, Cin, k, k] weight (+bias); k in {1, 3}, padding dilation * (k//2), stride in {1, 2}; dilation > 1 for 3x3
    stride-1 layers (the atrous layers of the segmentation net, reference lib/models/ctrnet/keypoint_seg_resnet.py:103-149)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, bias=True, dilation=1):
        super().__init__()
        assert kernel_size in (1, 3) and stride in (1, 2) and (dilation == 1 or (kernel_size == 3 and stride == 1))
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.dilation = kernel_size, stride, dilation
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, kernel_size, kernel_size))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        bound = 1.0 / math.sqrt(in_channels * kernel_size * kernel_size)
        nn.init.uniform_(self.weight, -bound, bound)
        if bias:
            nn.init.uniform_(self.bias, -bound, bound)

    def emit(self, pb, x, want_stats=False):
        if self.kernel_size == 1 and self.stride == 1 and not want_stats and x.H == 1 and x.W == 1 and x.dtype == torch.float32:
            # a 1x1 conv on pooled fp32 features is nn.Linear (depth_layer, depth_net.py:121-123 / full_net.py:271-274): the skinny
            # GEMM kernels with their ordered reduction instead of the conv path's split-K atomics
            return pb.linear(x, self.weight, self.bias)
        return pb.conv(x, self.weight, self.bias, stride=self.stride, want_stats=want_stats, dilation=self.dilation)

    def extra_repr(self):
        return f"{self.in_channels}, {self.out_channels}, k={self.kernel_size}, s={self.stride}"


class BatchNorm2d(nn.BatchNorm2d, PlannedModule):
    """Parameter / buffer holder with torch's class in its bases: `isinstance(m, torch.nn.BatchNorm2d)` checks of the reference's
    trainers (scripts/train_sim2real.py:144-146 freezes BatchNorm that way) see these modules; it never runs torch's forward."""

    def __init__(self, num_features, momentum=0.1, eps=1e-5):
        nn.BatchNorm2d.__init__(self, num_features, eps=eps, momentum=momentum)      # (reaches PlannedModule.__init__ through the MRO)

    def forward(self, x):
        raise NotImplementedError("BatchNorm2d is always fused with its producing conv in a plan")


class BatchNorm1d(nn.BatchNorm1d, PlannedModule):
    """nn.BatchNorm1d of the add_fc heads (full_net.py:150-157, depth_net.py:44-70): the same parameters / buffers, planned as a
    BatchNorm over [N, 1, 1, C]."""

    def __init__(self, num_features, momentum=0.1, eps=1e-5):
        nn.BatchNorm1d.__init__(self, num_features, eps=eps, momentum=momentum)

    def forward(self, x):
        raise NotImplementedError("BatchNorm1d is always fused with its producing layer in a plan")


def conv_bn(pb, x, conv, bn):
    """conv output (raw) as a BN term; the conv epilogue gathers batch statistics in train mode."""
    return Term(conv.emit(pb, x, want_stats=pb.plan.training), bn)


def conv3x3(in_planes, out_planes, stride=1):
    return Conv2d(in_planes, out_planes, 3, stride=stride, bias=False)


class _Downsample(nn.Sequential):
    """conv1x1 + BN on the residual path (keys ``downsample.0`` / ``downsample.1``)."""

    def __init__(self, cin, cout, stride):
        super().__init__(Conv2d(cin, cout, 1, stride=stride, bias=False), BatchNorm2d(cout, momentum=BN_MOMENTUM))


class BasicBlock(SingleTensorModule):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = conv3x3(inplanes, planes, stride)
        self.bn1 = BatchNorm2d(planes, momentum=BN_MOMENTUM)
        self.conv2 = conv3x3(planes, planes)
        self.bn2 = BatchNorm2d(planes, momentum=BN_MOMENTUM)
        self.downsample = downsample
        self.stride = stride

    def emit(self, pb, x):
        if self.downsample is None and self.stride == 1:
            # inference plans, 32 / 64-channel branch blocks: the whole block as one launch, the intermediate in LDS
            out = pb.basic_block_eval(x, self.conv1.weight, self.bn1, self.conv2.weight, self.bn2)
            if out is not None:
                return out
            # train-mode branch blocks of the row-strip shapes: BatchNorm + ReLU of the interior inside the convolutions
            out = pb.conv_bn_relu_conv(x, self.conv1.weight, self.bn1, self.conv2.weight, self.bn2)
            if out is not None:
                return out
        h = pb.act([conv_bn(pb, x, self.conv1, self.bn1)], relu=True)
        skip = Term(x) if self.downsample is None else conv_bn(pb, x, self.downsample[0], self.downsample[1])
        return pb.act([conv_bn(pb, h, self.conv2, self.bn2), skip], relu=True)


class Bottleneck(SingleTensorModule):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = BatchNorm2d(planes, momentum=BN_MOMENTUM)
        self.conv2 = Conv2d(planes, planes, 3, stride=stride, bias=False)
        self.bn2 = BatchNorm2d(planes, momentum=BN_MOMENTUM)
        self.conv3 = Conv2d(planes, planes * self.expansion, 1, bias=False)
        self.bn3 = BatchNorm2d(planes * self.expansion, momentum=BN_MOMENTUM)
        self.downsample = downsample
        self.stride = stride

    def emit(self, pb, x):
        h = pb.act([conv_bn(pb, x, self.conv1, self.bn1)], relu=True)
        h = pb.act([conv_bn(pb, h, self.conv2, self.bn2)], relu=True)
        if self.stride == 1:
            # training plans, the wide high-resolution blocks (layer1, the first incre-module): conv3's 134 MB output - and the raw output
            # of a 1x1 projection shortcut - are never stored, the BatchNorm + shortcut + ReLU and their backward run inside pointwise
            # launches that multiply again (PlanBuilder.bottleneck_tail)
            proj = None if self.downsample is None else (self.downsample[0].weight, self.downsample[1])
            out = pb.bottleneck_tail(h, self.conv3.weight, self.bn3, x, proj=proj)
            if out is not None:
                return out
        skip = Term(x) if self.downsample is None else conv_bn(pb, x, self.downsample[0], self.downsample[1])
        return pb.act([conv_bn(pb, h, self.conv3, self.bn3), skip], relu=True)


blocks_dict = {"BASIC": BasicBlock, "BOTTLENECK": Bottleneck}


def _block_stack(block, inplanes, planes, count, stride=1):
    """`count` blocks; the first gets a 1x1 projection when shape changes (HRnet.py:139-175, 431-465)."""
    ds = None
    if stride != 1 or inplanes != planes * block.expansion:
        ds = _Downsample(inplanes, planes * block.expansion, stride)
    mods = [block(inplanes, planes, stride, ds)]
    mods += [block(planes * block.expansion, planes) for _ in range(1, count)]
    return nn.Sequential(*mods)


def _emit_seq(pb, seq, x):
    for m in seq:
        x = m.emit(pb, x)
    return x


class HighResolutionModule(PlannedModule):
    """Parallel branches + all-to-all cross-resolution fuse (reference HRnet.py:101-265)."""

    def __init__(self, num_branches, blocks, num_blocks, num_inchannels, num_channels, fuse_method,
                 multi_scale_output=True):
        super().__init__()
        for name, lst in (("NUM_BLOCKS", num_blocks), ("NUM_CHANNELS", num_channels),
                          ("NUM_INCHANNELS", num_inchannels)):
            if num_branches != len(lst):
                msg = "NUM_BRANCHES({}) <> {}({})".format(num_branches, name, len(lst))
                logger.error(msg)
                raise ValueError(msg)
        self.num_inchannels = num_inchannels
        self.fuse_method = fuse_method
        self.num_branches = num_branches
        self.multi_scale_output = multi_scale_output
        branches = []
        for b in range(num_branches):
            branches.append(_block_stack(blocks, num_inchannels[b], num_channels[b], num_blocks[b]))
            self.num_inchannels[b] = num_channels[b] * blocks.expansion
        self.branches = nn.ModuleList(branches)
        self.fuse_layers = self._make_fuse_layers()

    def _make_fuse_layers(self):
        if self.num_branches == 1:
            return None
        ch = self.num_inchannels
        rows = []
        for i in range(self.num_branches if self.multi_scale_output else 1):
            row = []
            for j in range(self.num_branches):
                if j > i:      # lower resolution -> 1x1 conv + BN, upsampled in the fuse kernel
                    row.append(nn.Sequential(Conv2d(ch[j], ch[i], 1, bias=False), BatchNorm2d(ch[i])))
                elif j == i:
                    row.append(None)
                else:          # higher resolution -> (i-j) stride-2 3x3 convs
                    steps = []
                    for k in range(i - j):
                        cout = ch[i] if k == i - j - 1 else ch[j]
                        steps.append(nn.Sequential(Conv2d(ch[j], cout, 3, stride=2, bias=False), BatchNorm2d(cout)))
                    row.append(nn.Sequential(*steps))
            rows.append(nn.ModuleList(row))
        return nn.ModuleList(rows)

    def get_num_inchannels(self):
        return self.num_inchannels

    def emit_fuse(self, pb, xs):
        """All-to-all cross-resolution fuse of the branch outputs (HRnet.py:243-265).

        The paths j -> i are independent until the sums: they are emitted into one virtual lane per SOURCE branch j (all
        paths out of branch j accumulate into the gradient of xs[j], so they stay in one lane), the sums into one
        virtual lane per output - the lock-step merge turns same-shaped steps of different lanes into batched launches."""
        if self.num_branches == 1:
            return xs
        nb, rows = self.num_branches, self.fuse_layers
        terms = [[None] * nb for _ in rows]
        with pb.parallel(nb, virtual=True) as par:
            for j in range(nb):
                with par.lane(j):
                    for i in range(len(rows)):
                        if j > i:      # lower resolution: 1x1 conv + BN, upsampled inside the sum kernel
                            t = conv_bn(pb, xs[j], rows[i][j][0], rows[i][j][1])
                            t.up = 2 ** (j - i)
                            terms[i][j] = t
                    for i in range(len(rows)):
                        if j < i:      # higher resolution: (i - j) stride-2 3x3 steps
                            h = xs[j]
                            steps = list(rows[i][j])
                            for st in steps[:-1]:
                                h = pb.act([conv_bn(pb, h, st[0], st[1])], relu=True)
                            terms[i][j] = conv_bn(pb, h, steps[-1][0], steps[-1][1])
        outs = [None] * len(rows)
        with pb.parallel(len(rows), virtual=True) as par:
            for i in range(len(rows)):
                with par.lane(i):
                    terms[i][i] = Term(xs[i])
                    outs[i] = pb.act(terms[i], relu=True)
        return outs

    def emit(self, pb, xs, virtual=False):
        # the branches are independent until the fuse layers: one lane each (a HIP stream / graph branch, or - virtual -
        # a lane of the lock-step merge inside the caller's stream)
        xs = list(xs)
        with pb.parallel(self.num_branches, virtual=virtual) as par:
            for b in range(self.num_branches):
                with par.lane(b):
                    xs[b] = _emit_seq(pb, self.branches[b], xs[b])
        return self.emit_fuse(pb, xs)

    def forward(self, x):
        outs = self._run(*x)
        return list(outs)

    def _build(self, pb, *xs):
        names, hs, imgs = [], [], {}
        for i, x in enumerate(xs):
            N, Cc, H, W = x.shape
            t = pb.image_input(f"x{i}", N, Cc, H, W, u8=x.dtype == torch.uint8)
            t.requires_grad = pb.plan.need_grad and x.requires_grad
            names.append(f"x{i}")
            hs.append(t)
            imgs[f"x{i}"] = t
        ys = self.emit(pb, hs)
        outs = []
        for y in ys:
            holder = pb.nchw_output(y)
            holder["handle"] = y
            outs.append(("nchw", holder, None))
        return names, outs, imgs


def _transition(pb, tr, src):
    steps = [tr] if isinstance(tr[0], Conv2d) else list(tr)
    for st in steps:
        src = pb.act([conv_bn(pb, src, st[0], st[1])], relu=True)
    return src


# "nets": one chain (a stream in the plan's hybrid mode) per trunk, the branches of a module are virtual lanes merged
# into batched launches inside it (default).  "flat2": two streams per trunk - the high-resolution branch (HBM-heavy,
# small-K launches) and the other branches batched - 1.7 % faster on the step at the end of round 2 (39.7 against 40.4 ms on
# one box, inside the box-to-box spread) for 600 more launches and a less efficient conv family (27.9 against 25.4 ms of
# kernel time, roofline fraction 0.177 against 0.197): not the default.  "flat": round 1's structure - every branch of
# every trunk is a lane of one flat block.
TRUNK_LANES = os.environ.get("HRP_TRUNK_LANES", "nets")
# lanes of bf16 INFERENCE plans when HRP_TRUNK_LANES is "nets": "flat22" = two streams per trunk (branches {0, 1} | {2, 3}), "" = as
# the training plans
EVAL_LANES = "flat22"
# lanes of a plan with ONE trunk (RootNet alone: the metric's literal workload) when HRP_TRUNK_LANES is "nets": nothing else
# overlaps its chain of launches, so the high-resolution branch gets a stream of its own ("flat2": 21.75 -> 21.29 ms per step at
# B = 64; "flat22" 21.72, "flat" 22.49)
SINGLE_NET_LANES = "flat2"


def _trunk_segments(net):
    """Stem .. stage 4 of ONE net (reference HRnet.py:500-533) as four chain segments: emit_trunks joins the nets'
    streams between them, which gives the backward top-level positions (between stages) where the data-parallel step
    may split it, and lets the late weight pack join after the stem."""
    def stem(pb, x):
        h = pb.act([conv_bn(pb, x, net.conv1, net.bn1)], relu=True)
        h = pb.act([conv_bn(pb, h, net.conv2, net.bn2)], relu=True)
        h = _emit_seq(pb, net.layer1, h)
        return [h if tr is None else _transition(pb, tr, h) for tr in net.transition1]

    def stage(mods, trans, number=0):
        def seg(pb, ys):
            if TRUNK_FP32_FROM and not net.generate_hm and number == int(TRUNK_FP32_FROM):
                ys = [pb.cast(y, torch.float32) for y in ys]
            for m in mods:
                ys = m.emit(pb, ys, virtual=True)
            if trans is not None:
                # a new branch always starts from the LAST (lowest-resolution) output, HRnet.py:516-529
                ys = [ys[j] if tr is None else _transition(pb, tr, ys[-1]) for j, tr in enumerate(trans)]
            return ys
        return seg
    return [stem, stage(net.stage2, net.transition2, 2), stage(net.stage3, net.transition3, 3), stage(net.stage4, None, 4)]


def emit_trunks(pb, nets, xs, rider=None):
    """Trunks (stem .. stage4, reference HRnet.py:500-533) of one or more HRNets with the same stage layout,
    emitted in lockstep so that every independent chain of every net - stem, each branch of the current
    module, fuse + transition - gets its own lane inside ONE flat parallel block.  (Blocks are kept flat:
    forking a lane from a forked lane crashes HIP stream capture on ROCm 7.)  -> per net the list of the
    stage-4 branch outputs.

    `rider`: optional generator that emits an independent chain unit by unit (the ResNet regression trunk of the
    shipped full.yaml next to the HRNet root trunk); every parallel block gets one more lane that advances it by
    one unit, so the chain overlaps the HRNet branches without nesting blocks."""
    n = len(nets)
    mode = TRUNK_LANES
    if mode == "nets" and EVAL_LANES and pb.fuse_inference and pb.plan.dtype == torch.bfloat16:
        # inference plans with the fused BasicBlock launch (csrc/conv_block.h): the two high-resolution branches of a net (one
        # chip-exclusive launch per block) and its low-resolution branches (batched whole-image launches) on two streams
        mode = EVAL_LANES
    if mode == "nets" and n == 1 and rider is None and SINGLE_NET_LANES and pb.plan.dtype == torch.bfloat16:
        mode = SINGLE_NET_LANES
    if mode == "nets" and rider is None:
        ys = list(xs)
        segs = [_trunk_segments(net) for net in nets]
        for k in range(4):
            with pb.parallel(n) as par:
                for i in range(n):
                    with par.lane(i):
                        ys[i] = segs[i][k](pb, ys[i])
        return ys
    alive = [rider is not None]

    def ride(par, lane):
        if alive[0]:
            with par.lane(lane):
                try:
                    next(rider)
                except StopIteration:
                    alive[0] = False

    extra = lambda: 1 if alive[0] else 0
    stages = [(net.stage2, net.stage3, net.stage4) for net in nets]
    assert all(len(st[k]) == len(stages[0][k]) for st in stages for k in range(3)), "nets differ in stage layout"
    ys = [None] * n
    with pb.parallel(n + extra()) as par:
        ride(par, n)
        for i, (net, x) in enumerate(zip(nets, xs)):
            with par.lane(i):
                h = pb.act([conv_bn(pb, x, net.conv1, net.bn1)], relu=True)
                h = pb.act([conv_bn(pb, h, net.conv2, net.bn2)], relu=True)
                h = _emit_seq(pb, net.layer1, h)
                ys[i] = [h if tr is None else _transition(pb, tr, h) for tr in net.transition1]
    for k in range(3):
        for mi in range(len(stages[0][k])):
            mods = [st[k][mi] for st in stages]
            if mode == "flat22":
                # two lanes per net: branches {0, 1} and branches {2, 3}, each a lock-step merge of its branches
                with pb.parallel(2 * n + extra()) as par:
                    ride(par, 2 * n)
                    for i, m in enumerate(mods):
                        ys[i] = list(ys[i])
                        for half, bs in enumerate(([b for b in range(m.num_branches) if b < 2], [b for b in range(m.num_branches) if b >= 2])):
                            if not bs:
                                continue
                            with par.lane(2 * i + half):
                                with pb.parallel(len(bs), virtual=True) as vp:
                                    for q, b in enumerate(bs):
                                        with vp.lane(q):
                                            ys[i][b] = _emit_seq(pb, m.branches[b], ys[i][b])
            elif mode == "flat3":
                # two lanes per RESOLUTION CLASS: the high-resolution branch of every net (HBM-heavy, small-K launches:
                # the batch runs the light conv kernel with persistent workgroups) and all other branches of every net
                # (MFMA-heavy batched launches) - complementary kernels on two streams
                for i in range(n):
                    ys[i] = list(ys[i])
                with pb.parallel(2 + extra()) as par:
                    ride(par, 2)
                    with par.lane(0):
                        with pb.parallel(n, virtual=True) as vp:
                            for i, m in enumerate(mods):
                                with vp.lane(i):
                                    ys[i][0] = _emit_seq(pb, m.branches[0], ys[i][0])
                    with par.lane(1):
                        items = [(i, b) for i, m in enumerate(mods) for b in range(1, m.num_branches)]
                        with pb.parallel(len(items), virtual=True) as vp:
                            for q, (i, b) in enumerate(items):
                                with vp.lane(q):
                                    ys[i][b] = _emit_seq(pb, mods[i].branches[b], ys[i][b])
            elif mode == "flat2":
                # two lanes per net: the high-resolution branch (HBM-heavy launches) and the other branches as virtual
                # lanes of one chain (MFMA-heavy batched launches)
                with pb.parallel(2 * n + extra()) as par:
                    ride(par, 2 * n)
                    for i, m in enumerate(mods):
                        ys[i] = list(ys[i])
                        with par.lane(2 * i):
                            ys[i][0] = _emit_seq(pb, m.branches[0], ys[i][0])
                        with par.lane(2 * i + 1):
                            with pb.parallel(m.num_branches - 1, virtual=True) as vp:
                                for b in range(1, m.num_branches):
                                    with vp.lane(b - 1):
                                        ys[i][b] = _emit_seq(pb, m.branches[b], ys[i][b])
            else:
              nl = sum(m.num_branches for m in mods)
              with pb.parallel(nl + extra()) as par:
                ride(par, nl)
                lane = 0
                for i, m in enumerate(mods):
                    ys[i] = list(ys[i])
                    for b in range(m.num_branches):
                        with par.lane(lane):
                            ys[i][b] = _emit_seq(pb, m.branches[b], ys[i][b])
                        lane += 1
            last = mi == len(stages[0][k]) - 1
            with pb.parallel(n + extra()) as par:
                ride(par, n)
                for i, (net, m) in enumerate(zip(nets, mods)):
                    with par.lane(i):
                        ys[i] = m.emit_fuse(pb, ys[i])
                        trans = (net.transition2, net.transition3, None)[k]
                        if last and trans is not None:
                            # a new branch always starts from the LAST (lowest-resolution) output, HRnet.py:516-529
                            ys[i] = [ys[i][j] if tr is None else _transition(pb, tr, ys[i][-1])
                                     for j, tr in enumerate(trans)]
    return ys


class PoseHighResolutionNet(PlannedModule):
    """HRNet trunk with optional heat-map conv and 2048-d classification-head feature
    (reference HRnet.py:274-570)."""

    def __init__(self, cfg, **kwargs):
        super().__init__()
        extra = cfg["MODEL"]["EXTRA"]
        self.generate_feat = kwargs["generate_feat"]
        self.generate_hm = kwargs.get("generate_hm", True)
        self.conv1 = Conv2d(3, 64, 3, stride=2, bias=False)
        self.bn1 = BatchNorm2d(64, momentum=BN_MOMENTUM)
        self.conv2 = Conv2d(64, 64, 3, stride=2, bias=False)
        self.bn2 = BatchNorm2d(64, momentum=BN_MOMENTUM)
        self.layer1 = _block_stack(Bottleneck, 64, 64, 4)
        pre = [256]
        for idx, name in enumerate(("STAGE2", "STAGE3", "STAGE4")):
            scfg = extra[name]
            setattr(self, name.lower() + "_cfg", scfg)
            block = blocks_dict[scfg["BLOCK"]]
            chans = [c * block.expansion for c in scfg["NUM_CHANNELS"]]
            setattr(self, f"transition{idx + 1}", self._make_transition_layer(pre, chans))
            mso = True if name != "STAGE4" else self.generate_feat
            stage, pre = self._make_stage(scfg, chans, multi_scale_output=mso)
            setattr(self, name.lower(), stage)
        if self.generate_feat:
            self.incre_modules, self.downsamp_modules, self.final_feat_layer = self._make_cls_head(pre)
        if self.generate_hm:
            k = extra["FINAL_CONV_KERNEL"]
            self.final_layer = Conv2d(pre[0], cfg["MODEL"]["NUM_JOINTS"] * cfg["MODEL"]["DEPTH_DIM"], k, bias=True)
        self.pretrained_layers = extra["PRETRAINED_LAYERS"]

    # -- construction helpers (key layout of HRnet.py:341-429) -----------------------------------------
    def _make_cls_head(self, pre_stage_channels):
        head_channels = [32, 64, 128, 256]
        incre = nn.ModuleList([_block_stack(Bottleneck, c, head_channels[i], 1)
                               for i, c in enumerate(pre_stage_channels)])
        down = []
        for i in range(len(pre_stage_channels) - 1):
            cin, cout = head_channels[i] * 4, head_channels[i + 1] * 4
            down.append(nn.Sequential(Conv2d(cin, cout, 3, stride=2, bias=True),
                                      BatchNorm2d(cout, momentum=BN_MOMENTUM)))
        final = nn.Sequential(Conv2d(head_channels[3] * 4, 2048, 1, bias=True),
                              BatchNorm2d(2048, momentum=BN_MOMENTUM))
        return incre, nn.ModuleList(down), final

    def _make_transition_layer(self, pre, cur):
        layers = []
        for i in range(len(cur)):
            if i < len(pre):
                if cur[i] != pre[i]:
                    layers.append(nn.Sequential(Conv2d(pre[i], cur[i], 3, bias=False), BatchNorm2d(cur[i])))
                else:
                    layers.append(None)
            else:
                steps = []
                for j in range(i + 1 - len(pre)):
                    cout = cur[i] if j == i - len(pre) else pre[-1]
                    steps.append(nn.Sequential(Conv2d(pre[-1], cout, 3, stride=2, bias=False), BatchNorm2d(cout)))
                layers.append(nn.Sequential(*steps))
        return nn.ModuleList(layers)

    def _make_stage(self, layer_config, num_inchannels, multi_scale_output=True):
        block = blocks_dict[layer_config["BLOCK"]]
        n = layer_config["NUM_MODULES"]
        mods = []
        for i in range(n):
            mso = multi_scale_output or i != n - 1
            mods.append(HighResolutionModule(layer_config["NUM_BRANCHES"], block, layer_config["NUM_BLOCKS"],
                                             num_inchannels, layer_config["NUM_CHANNELS"],
                                             layer_config["FUSE_METHOD"], mso))
            num_inchannels = mods[-1].get_num_inchannels()
        return nn.Sequential(*mods), num_inchannels

    # -- plan description ---------------------------------------------------------------------------------
    def emit_trunk(self, pb, x):
        return emit_trunks(pb, [self], [x])[0]

    def emit_heads(self, pb, ys, feat_out=None):
        """-> (heat-map tensor or None, fp32 feature tensor or None)."""
        heat = feat = None
        if self.generate_hm:
            heat = self.final_layer.emit(pb, ys[0])
        if self.generate_feat:
            if HEAD_FP32 and not self.generate_hm and ys[0].dtype != torch.float32:
                # the classification head of a feature-only trunk (the DepthNet) in fp32: its pooled feature sets the root depth,
                # where 1 mm is 3 px for a key-point 0.13 m in front of the camera
                ys = [pb.cast(y, torch.float32) for y in ys]
            # HRnet.py:535-541 interleaves incre_modules[i + 1] with the down-sampling chain; the four incre modules depend on the
            # trunk's outputs only, so they are emitted side by side in virtual lanes: the 1 x 1 / 3 x 3 / 1 x 1 + projection launches
            # of the three low-resolution Bottlenecks batch position by position (each ran alone for 15-25 us at < 1 TB/s)
            incs = [None] * len(self.incre_modules)
            if HEAD_INCRE_LANES:
                with pb.parallel(len(incs), virtual=True) as par:
                    for i in range(len(incs)):
                        with par.lane(i):
                            incs[i] = self.incre_modules[i][0].emit(pb, ys[i])
            y = incs[0] if incs[0] is not None else self.incre_modules[0][0].emit(pb, ys[0])
            for i, dm in enumerate(self.downsamp_modules):
                d = pb.act([conv_bn(pb, y, dm[0], dm[1])], relu=True)
                inc = incs[i + 1] if incs[i + 1] is not None else self.incre_modules[i + 1][0].emit(pb, ys[i + 1])
                y = pb.act([Term(inc), Term(d)], relu=False)
            y = pb.act([conv_bn(pb, y, self.final_feat_layer[0], self.final_feat_layer[1])], relu=True)
            feat = pb.avgpool(y, out=feat_out)
        return heat, feat

    def emit(self, pb, x, feat_out=None):
        return self.emit_heads(pb, self.emit_trunk(pb, x), feat_out=feat_out)

    def _build(self, pb, x):
        N, Cc, H, W = x.shape
        t = pb.image_input("x", N, Cc, H, W, u8=x.dtype == torch.uint8)
        t.requires_grad = pb.plan.need_grad and x.requires_grad
        heat, feat = self.emit(pb, t)
        outs = []
        if heat is not None:
            holder = pb.nchw_output(heat)
            holder["handle"] = heat
            outs.append(("nchw", holder, None))
        if feat is not None:
            outs.append(("dense", feat, (N, feat.C)))
        return ["x"], outs, {"x": t}

    def forward(self, x):
        outs = self._run(x)
        if self.generate_hm and self.generate_feat:
            return outs[0], outs[1]
        return outs[0]

    def init_weights(self, pretrained=""):
        logger.info("=> init weights from normal distribution")
        for m in self.modules():
            if isinstance(m, Conv2d):
                nn.init.normal_(m.weight, std=0.001)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        if os.path.isfile(pretrained):
            print(f"Loading hrnet pretrained weights (ImageNet) from {pretrained}")
            state = torch.load(pretrained, map_location="cpu")
            keep = {k: v for k, v in state.items()
                    if k.split(".")[0] in self.pretrained_layers or self.pretrained_layers[0] == "*"}
            self.load_state_dict(keep, strict=False)
        elif pretrained:
            logger.error("=> please download pre-trained models first!")
            raise ValueError("{} is not exist!".format(pretrained))


def load_hrnet_cfg(file_name):
    """YAML file -> attribute dict; falls back to the built-in W32/W48 tables when the file is absent."""
    if os.path.isfile(file_name):
        import yaml
        with open(file_name) as f:
            return AttrDict(yaml.load(f, Loader=yaml.FullLoader))
    key = os.path.splitext(os.path.basename(file_name))[0]
    if key not in HRNET_CONFIGS:
        raise FileNotFoundError(file_name)
    return AttrDict(HRNET_CONFIGS[key])


def get_hrnet(type_name, num_joints, depth_dim, pretrain=True, **kwargs):
    cfg = load_hrnet_cfg(f"./lib/models/backbones/configs/hrnet_w{type_name}.yaml")
    cfg["MODEL"]["NUM_JOINTS"] = num_joints
    cfg["MODEL"]["DEPTH_DIM"] = depth_dim
    model = PoseHighResolutionNet(cfg, **kwargs)
    if pretrain:
        pre = cfg["MODEL"]["PRETRAINED"]
        model.init_weights(pretrained=pre if os.path.isfile(pre) else "")
    return model
