"""``seg_mask_inference`` (reference lib/models/ctrnet/mask_inference.py:12-57), the mask network of the self-supervised trainer
(scripts/train_sim2real.py:89, 412): raw images [B, 3, 480, 640] with values 0 .. 255 -> foreground probability [B, 1, 240, 320].

The reference moves every image to the host, resizes it with PIL, normalises it with torchvision transforms, uploads the batch
and runs CtRNet's DeepLabv3-ResNet50.  Here the whole call is ONE static plan on the device: hrp_pil_resize_normalize (bit-exact
with PIL.Image.resize on the bytes) -> the network (KeyPointSegNet.emit_logits) -> bilinear up-sampling + sigmoid in one launch.
Same constructor, attribute names (``net``, ``args``) and state-dict keys (``net.keypoint_seg_predictor.module.*``).
PARITY: the resize is pinned against Pillow itself; the network is parity-unpinned (keypoint_seg_resnet.py docstring)."""
import argparse

import torch

from hrpe_amd.runtime import PlannedModule
from .CtRNet import CtRNet

MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)          # transforms.Normalize of mask_inference.py:20


class seg_mask_inference(PlannedModule):
    def __init__(self, intrinsics, dataset, image_hw=(480, 640), scale=0.5, allow_random_init=False):
        """allow_random_init: keep the random initialisation when the checkpoint of `dataset` does not exist (tests and benchmarks on
        synthetic weights); by default a missing file raises FileNotFoundError as the reference's torch.load does (CtRNet.py:35)."""
        super().__init__()
        self.args = self.set_args(intrinsics, dataset, image_hw, scale)
        self.args.allow_random_seg_init = bool(allow_random_init)
        self.net = CtRNet(self.args)
        for p in self.parameters():          # never trained (train_sim2real.py:412 detaches the output; the optimizer holds the
            p.requires_grad_(False)          # pose network's parameters only): plans of this module are inference plans
        PlannedModule.train(self, False)

    def train(self, mode=True):
        return self                          # (CtRNet.py:30 keeps the predictor in eval(); there is no train mode of this network)

    def set_args(self, intrinsics, dataset, image_hw=(480, 640), scale=0.5):
        args = argparse.ArgumentParser().parse_args("")
        args.use_gpu = True
        args.robot_name = "Panda"
        args.n_kp = 7
        args.scale = scale
        args.height, args.width = image_hw
        args.fx, args.fy, args.px, args.py = intrinsics
        args.width, args.height = int(args.width * args.scale), int(args.height * args.scale)
        args.fx, args.fy, args.px, args.py = args.fx * args.scale, args.fy * args.scale, args.px * args.scale, args.py * args.scale
        name = next((k for k in ("realsense", "azure", "kinect", "orb") if k in dataset), "azure")
        args.keypoint_seg_model_path = f"models/panda_segmentation/{name}.pth"
        return args

    def _build(self, pb, img):
        if pb.plan.need_grad:
            raise NotImplementedError("seg_mask_inference: no gradients (call it as the trainer does: the output is detached)")
        N, Cc, H, W = img.shape
        assert Cc == 3
        t = pb.pil_resize_input("img", N, H, W, self.args.scale, MEAN, STD)
        seg = self.net.keypoint_seg_predictor.module
        logits = seg.emit_logits(pb, t)
        Ho, Wo = int(H * self.args.scale), int(W * self.args.scale)
        holder = pb.bilinear_nchw_output(logits, Ho, Wo, sigmoid=True)
        holder["handle"] = logits
        return ["img"], [("nchw", holder, None)], {"img": t}

    def forward(self, img_tensor):
        return self._run(img_tensor)[0]
