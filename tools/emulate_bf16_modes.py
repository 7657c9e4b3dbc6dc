#!/usr/bin/env python3
"""Development tool (CPU, no GPU needed): WHERE does the bf16 DepthNet lose key-point 0?  

Replays the eval-mode forward of the full network on the reference's eval fixture (tests/golden/golden_full_eval.npz inputs,
synthesised weights) with torch on the CPU, emulating the rounding points of the HIP inference plan in the DepthNet trunk
(rootnet_backbone): bf16 weights, bf16 conv inputs, fp32 accumulation, folded BatchNorm scale / shift + residual + ReLU in fp32,
ONE rounding when a tensor is stored.  Modes:

  fp32      nothing rounded (the oracle itself)
  bf16      every stored activation is bf16 (what the library's bf16 plans do)
  res32     the RESIDUAL STREAM stays fp32: block outputs, fuse sums and transition outputs are stored in fp32 (the convolution
            that consumes them reads a bf16-rounded copy); block interiors (conv1 -> conv2, bottleneck interiors) stay bf16
  res32+w   as res32, and the weights stay fp32 too (isolates activation rounding from weight rounding)
  w_only    fp32 activations, bf16 weights

Prints per mode the root depth error and the end-to-end key-point error in pixels against the reference run in float64
(golden_full_eval_fp64.npz), the figure bench.py reports as max_px_err.*_vs_fp64.  The regression trunk runs in fp32 in every
mode: key-point 0's error is the DepthNet's.
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

import oracle.heads as oheads  # noqa: E402
import oracle.hrnet as ohr  # noqa: E402
from oracle import fk as ofk  # noqa: E402
from synth import synth_inputs, synth_state_dict  # noqa: E402


def q(t):
    return t.bfloat16().float()


class Emu:
    """hrnet_w32_forward (feature head only, eval) with explicit storage precision."""

    def __init__(self, sd, prefix, mode):
        self.sd, self.p, self.mode = sd, prefix, mode
        self.act = (lambda t: t) if mode in ("fp32", "w_only") else q          # interior activations
        self.stream = (lambda t: t) if mode in ("fp32", "w_only", "res32", "res32+w") else q   # residual stream / sums
        self.wq = q if mode in ("bf16", "res32", "w_only") else (lambda t: t)

    def conv_bn(self, x, ck, bk, stride=1):
        """conv (bf16 operands, fp32 accumulate) with the BatchNorm folded into a per-channel scale / shift (fp32)."""
        sd, p = self.sd, self.p
        w = self.wq(sd[p + ck + ".weight"])
        b = sd.get(p + ck + ".bias")
        xin = x if self.mode in ("fp32", "w_only") else q(x)         # the conv reads a bf16 copy of whatever it is given
        y = F.conv2d(xin, w, None, stride=stride, padding=w.shape[-1] // 2)
        g, be = sd[p + bk + ".weight"], sd[p + bk + ".bias"]
        m, v = sd[p + bk + ".running_mean"], sd[p + bk + ".running_var"]
        sc = g / torch.sqrt(v + ohr.BN_EPS)
        sh = be - m * sc
        if b is not None:
            sh = sh + b * sc
        return y * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)

    def basic(self, key, x):
        h = self.act(F.relu(self.conv_bn(x, key + ".conv1", key + ".bn1")))
        return self.stream(F.relu(self.conv_bn(h, key + ".conv2", key + ".bn2") + x))

    def bottleneck(self, key, x):
        h = self.act(F.relu(self.conv_bn(x, key + ".conv1", key + ".bn1")))
        h = self.act(F.relu(self.conv_bn(h, key + ".conv2", key + ".bn2")))
        o = self.conv_bn(h, key + ".conv3", key + ".bn3")
        if (self.p + key + ".downsample.0.weight") in self.sd:
            x = self.conv_bn(x, key + ".downsample.0", key + ".downsample.1")
        return self.stream(F.relu(o + x))

    def module(self, key, xs, mso=True):
        nb = len(xs)
        xs = list(xs)
        for b in range(nb):
            for k in range(ohr.BLOCKS_PER_BRANCH):
                xs[b] = self.basic(f"{key}.branches.{b}.{k}", xs[b])
        outs = []
        for i in range(nb if mso else 1):
            y = None
            for j in range(nb):
                fk_ = f"{key}.fuse_layers.{i}.{j}"
                if j == i:
                    t = xs[j]
                elif j > i:
                    t = self.act(self.conv_bn(xs[j], fk_ + ".0", fk_ + ".1"))      # a path's output is a stored tensor
                    t = F.interpolate(t, scale_factor=2 ** (j - i), mode="nearest")
                else:
                    t = xs[j]
                    for k in range(i - j):
                        t = self.conv_bn(t, f"{fk_}.{k}.0", f"{fk_}.{k}.1", stride=2)
                        if k != i - j - 1:
                            t = F.relu(t)
                        t = self.act(t)
                y = t if y is None else y + t
            outs.append(self.stream(F.relu(y)))
        return outs

    def forward(self, x):
        st = self.stream
        x = self.act(F.relu(self.conv_bn(x, "conv1", "bn1", 2)))
        x = st(F.relu(self.conv_bn(x, "conv2", "bn2", 2)))
        for k in range(4):
            x = self.bottleneck(f"layer1.{k}", x)
        ys = [st(F.relu(self.conv_bn(x, "transition1.0.0", "transition1.0.1"))),
              st(F.relu(self.conv_bn(x, "transition1.1.0.0", "transition1.1.0.1", 2)))]
        ys = self.module("stage2.0", ys)
        ys = ys + [st(F.relu(self.conv_bn(ys[-1], "transition2.2.0.0", "transition2.2.0.1", 2)))]
        for m in range(4):
            ys = self.module(f"stage3.{m}", ys)
        ys = ys + [st(F.relu(self.conv_bn(ys[-1], "transition3.3.0.0", "transition3.3.0.1", 2)))]
        for m in range(3):
            ys = self.module(f"stage4.{m}", ys)
        y = self.bottleneck("incre_modules.0.0", ys[0])
        for i in range(3):
            d = self.act(F.relu(self.conv_bn(y, f"downsamp_modules.{i}.0", f"downsamp_modules.{i}.1", 2)))
            y = st(self.bottleneck(f"incre_modules.{i + 1}.0", ys[i + 1]) + d)
        y = F.relu(self.conv_bn(y, "final_feat_layer.0", "final_feat_layer.1"))
        return F.avg_pool2d(y, kernel_size=y.shape[2:]).view(y.shape[0], -1)


def main():
    import bench
    from hrpe_amd.lib.models.full_net import RootNetwithRegInt
    from hrpe_amd.lib.dataset.const import INITIAL_JOINT_ANGLE
    gdir = os.path.join(ROOT, "tests", "golden")
    g64 = np.load(os.path.join(gdir, "golden_full_eval_fp64.npz"))
    init = {"robot_type": "panda", "pose_params": INITIAL_JOINT_ANGLE, "cam_params": np.eye(4), "init_pose_from_mean": True}
    m = RootNetwithRegInt(init, bench.model_args(0.0))
    sd = synth_state_dict(m.state_dict())
    x_reg, x_root, kv, K = synth_inputs(2)
    robot = ofk.Robot(os.path.join(ROOT, "holistic-robot-pose-estimation_amd", "assets", "panda_kinematics.urdf"))
    uv64 = ofk.project(K.double(), torch.tensor(g64["xyz_fk"]))
    d64 = torch.tensor(g64["depth"]).double().view(-1) if "depth" in g64.files else None
    orig = ohr.hrnet_w32_forward
    torch.set_num_threads(8)
    print(f"{'mode':10s} {'|d feat| / |feat|':>18s} {'depth err [mm]':>16s}  px error per key-point (max over the 2 images)")
    feat_ref = None
    for mode in ("fp32", "bf16", "res32", "res32+w", "w_only"):
        holder = {}

        def patched(sd_, x, prefix="", generate_hm=True, generate_feat=True, training=False, taps=None):
            if prefix == "rootnet_backbone." and not generate_hm:
                f = Emu(sd_, prefix, mode).forward(x)
                holder["feat"] = f
                return f
            return orig(sd_, x, prefix=prefix, generate_hm=generate_hm, generate_feat=generate_feat, training=training, taps=taps)
        oheads.hrnet_w32_forward = patched
        with torch.no_grad():
            out = oheads.full_forward(sd, robot, x_reg, x_root, kv, K, training=False)
        oheads.hrnet_w32_forward = orig
        if feat_ref is None:
            feat_ref = holder["feat"]
        fe = float((holder["feat"] - feat_ref).norm() / feat_ref.norm())
        uv = ofk.project(K, out[7])
        e = (uv.double() - uv64).abs().amax(-1)           # [B, key-points]
        de = (out[4].double().view(-1) - d64).abs().max().item() * 1e3 if d64 is not None else float("nan")
        print(f"{mode:10s} {fe:18.2e} {de:16.4f}  " + " ".join(f"{v:7.3f}" for v in e.amax(0).tolist()))


if __name__ == "__main__":
    main()
