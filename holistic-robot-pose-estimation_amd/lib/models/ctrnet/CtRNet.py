"""``CtRNet`` as far as the mask network uses it (reference lib/models/ctrnet/CtRNet.py:10-45, 102-111): the key-point /
segmentation predictor behind a DataParallel-shaped wrapper (the reference's checkpoints carry the ``module.`` prefix) and
``inference_batch_images_onlyseg``.  The BPnP pose solver (CtRNet.py:47-100: OpenCV on the host) is out of scope."""
import os
import sys

import numpy as np
import torch

from .keypoint_seg_resnet import KeyPointSegNet


class _DataParallelShape(torch.nn.Module):
    """``torch.nn.DataParallel(model, device_ids=[0])`` as a name: one process per GPU here, nothing to scatter."""

    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, *a, **k):
        return self.module(*a, **k)


class CtRNet(torch.nn.Module):
    def __init__(self, args):
        super().__init__()
        self.args = args
        self.device = "cuda" if args.use_gpu else "cpu"
        self.keypoint_seg_predictor = _DataParallelShape(KeyPointSegNet(args, use_gpu=args.use_gpu))
        path = args.keypoint_seg_model_path
        if path is not None and os.path.exists(path):
            print("Loading keypoint segmentation model from {}".format(path))
            self.keypoint_seg_predictor.load_state_dict(torch.load(path, map_location="cpu"))
        elif path is not None and not getattr(args, "allow_random_seg_init", False):
            # the reference fails in torch.load (CtRNet.py:35): a relative default path (models/panda_segmentation/*.pth) resolved from
            # the wrong working directory must not leave train_sim2real self-training against random masks.  Tests and benchmarks on
            # synthetic weights opt in with args.allow_random_seg_init = True.
            raise FileNotFoundError(f"CtRNet: keypoint_seg_model_path {path!r} does not exist (set args.allow_random_seg_init = True to "
                                    "keep the random initialisation on purpose)")
        if args.use_gpu and torch.cuda.is_available():       # (the reference moves unconditionally; a build container has no GPU)
            self.keypoint_seg_predictor = self.keypoint_seg_predictor.cuda()
        self.keypoint_seg_predictor.eval()
        self.intrinsics = np.array([[args.fx, 0., args.px], [0., args.fy, args.py], [0., 0., 1.]])
        self.K = torch.tensor(self.intrinsics, device=self.device if torch.cuda.is_available() else "cpu", dtype=torch.float)

    def inference_batch_images_onlyseg(self, img):
        """img [B, 3, H, W] (normalised) -> foreground probability [B, 1, H, W] (CtRNet.py:102-111)"""
        _, segmentation = self.keypoint_seg_predictor(img)
        return torch.sigmoid(segmentation)

    def inference_single_image(self, img, joint_angles):
        raise NotImplementedError("CtRNet's BPnP pose solver (OpenCV on the host) is outside this library's path")

    inference_batch_images = inference_batch_images_seg_kp = inference_single_image
