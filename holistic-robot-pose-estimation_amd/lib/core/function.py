"""Caller-side step function pieces the benchmark / tests need (reference lib/core/function.py).

The model call itself is ``model(reg_images, root_images, k_values, K=other_K)`` exactly as in
function.py:119-120.  What lives here is the caller contract around it, restated for device tensors:
``k_values`` (function.py:88-98) without the per-sample Python loop and the loss assembly of
function.py:191-322 for the ``configs/panda/full.yaml`` choice of loss functions.  These are O(B*7*3)
element tensor expressions (torch ops, not part of the accelerated path); the two key-point
projections go through the projection kernel."""
import torch

from hrpe_amd.lib.utils.transforms import point_projection_from_3d_tensor

FULL_YAML_WEIGHTS = dict(pose=1.0, rot=1.0, trans=1.0, depth=10.0, uv=1.0, kp2d=10.0, kp3d=10.0,
                         kp2d_int=10.0, kp3d_int=10.0, align_3d=0.0)   # configs/panda/full.yaml:57-66


def compute_k_values(fx, fy, bboxes, real_bbox=(1000.0, 1000.0)):
    """k = sqrt(fx*fy*1000*1000 / max(|x2-x1|, |y2-y1|)^2)   (function.py:88-98)."""
    area = torch.max(torch.abs(bboxes[:, 2] - bboxes[:, 0]), torch.abs(bboxes[:, 3] - bboxes[:, 1])) ** 2
    return torch.sqrt(fx * fy * real_bbox[0] * real_bbox[1] / area).to(torch.float32)


def full_loss(pred, gt, K, root=3, image_size=256.0, weights=FULL_YAML_WEIGHTS):
    """pred: the model's 8-tuple.  gt: dict(pose, root_rot, root_trans, root_uv, kp3d, kp2d, mask).
    Returns (loss, dict of the ten terms named as in function.py:313-319)."""
    pose, rot, trans, root_uv, depth, uvd, xyz_int, xyz_fk = pred
    uv_int = point_projection_from_3d_tensor(K, xyz_int)        # function.py:121
    uv_fk = point_projection_from_3d_tensor(K, xyz_fk)          # function.py:122
    m = gt["mask"]
    t = {}
    t["loss_joint"] = torch.nn.functional.mse_loss(pose, gt["pose"])
    t["loss_rot"] = torch.nn.functional.mse_loss(rot, gt["root_rot"])
    t["loss_depth"] = torch.nn.functional.l1_loss(depth, gt["root_trans"][:, 2:3])
    e = torch.norm((root_uv - gt["root_uv"]) / image_size, dim=1) * m[:, root]
    t["loss_uv"] = e.sum() / (m[:, root] != 0).sum()
    e = torch.norm(trans - gt["root_trans"], dim=1)
    coeff = torch.where(e.mean() > 0.5, torch.exp(-20.0 * e).detach(), torch.ones_like(e))  # function.py:245-251
    t["loss_trans"] = (e * coeff).mean()
    t["loss_error3d"] = torch.norm(xyz_fk - gt["kp3d"], dim=2).mean()
    gt2d = gt["kp2d"] / image_size
    nvalid = (m != 0).sum()
    t["loss_error2d"] = (torch.norm(uv_fk / image_size - gt2d, dim=2) * m).sum() / nvalid
    t["loss_error3d_int"] = torch.norm(xyz_int - gt["kp3d"], dim=2).mean()
    t["loss_error2d_int"] = (torch.norm(uv_int / image_size - gt2d, dim=2) * m).sum() / nvalid
    t["loss_error3d_align"] = torch.norm(xyz_fk - xyz_int, dim=2).mean()
    w = weights
    loss = (w["pose"] * t["loss_joint"] + w["rot"] * t["loss_rot"] + w["uv"] * t["loss_uv"]
            + w["depth"] * t["loss_depth"] + w["trans"] * t["loss_trans"] + w["kp2d"] * t["loss_error2d"]
            + w["kp3d"] * t["loss_error3d"] + w["kp2d_int"] * t["loss_error2d_int"]
            + w["kp3d_int"] * t["loss_error3d_int"] + w["align_3d"] * t["loss_error3d_align"])
    return loss, t
