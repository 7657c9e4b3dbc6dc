"""Caller-side step function pieces the benchmark / tests need (reference lib/core/function.py).

The model call itself is ``model(reg_images, root_images, k_values, K=other_K)`` exactly as in
function.py:119-120.  What lives here is the caller contract around it, restated for device tensors:
``k_values`` (function.py:88-98) without the per-sample Python loop and the loss assembly of
function.py:191-322 for the ``configs/panda/full.yaml`` choice of loss functions.  These are O(B*7*3)
element tensor expressions (torch ops, not part of the accelerated path); the two key-point
projections go through the projection kernel."""
import torch

from hrpe_amd.lib.dataset.const import JOINT_NAMES
from hrpe_amd.lib.utils.geometries import rotmat_to_quat, rotmat_to_rot6d
from hrpe_amd.lib.utils.transforms import point_projection_from_3d_tensor

FULL_YAML_WEIGHTS = dict(pose=1.0, rot=1.0, trans=1.0, depth=10.0, uv=1.0, kp2d=10.0, kp3d=10.0,
                         kp2d_int=10.0, kp3d_int=10.0, align_3d=0.0)   # configs/panda/full.yaml:57-66


def compute_k_values(fx, fy, bboxes, real_bbox=(1000.0, 1000.0)):
    """k = sqrt(fx*fy*1000*1000 / max(|x2-x1|, |y2-y1|)^2)   (function.py:88-98)."""
    area = torch.max(torch.abs(bboxes[:, 2] - bboxes[:, 0]), torch.abs(bboxes[:, 3] - bboxes[:, 1])) ** 2
    return torch.sqrt(fx * fy * real_bbox[0] * real_bbox[1] / area).to(torch.float32)


def prepare_batch(input_batch, robot, device, reference_keypoint_id=3, use_origin_bbox=False, use_extended_bbox=True,
                  synthetic=True, rotation_dim=6):
    """The batch unpacking of function.py:25-98 for a DreamDataset batch (lib/dataset/dream.py:393-413), without the
    per-sample Python loops (gt pose/rot/trans at :50-64, k_values at :98) and without the fp32 image round trip:
    uint8 images stay uint8 on the way to the device (4x less PCIe traffic) and the model's input kernel does the
    ``.float() / 255.`` of :26,29 while it writes the trunk's NHWC layout (hrp_u8_nchw_to_nhwc); float images (the
    reference's loaders hand over float tensors holding 0..255) are scaled here as the reference does.

    Returns dict(reg_images, root_images, root_K, other_K, k_values, gt = dict(pose, rot, trans, root_rot,
    root_trans, root_depth, root_uv, kp3d, kp2d, mask)); ``gt`` feeds ``full_loss``.  `synthetic=False` (real
    datasets) needs the reference's BPnP solve on the CPU (function.py:66-74), which is outside this build."""
    if not synthetic:
        raise NotImplementedError("BPnP ground-truth rotation for real datasets (function.py:66-74) is not part of this build")

    def dev(t, dtype=torch.float32):
        return torch.as_tensor(t).to(device=device, dtype=dtype, non_blocking=True)

    def images(t):
        t = torch.as_tensor(t)
        if t.dtype == torch.uint8:
            return t.to(device, non_blocking=True)
        return dev(t) / 255.

    root, other = input_batch["root"], input_batch["other"]
    out = dict(root_images=images(root["images"]), reg_images=images(other["images"]),
               root_K=dev(root["K"]), other_K=dev(other["K"]))
    TCO = dev(input_batch["TCO"])
    jp = input_batch["jointpose"]
    pose = torch.stack([dev(jp[k]) for k in JOINT_NAMES[robot.robot_type]], dim=1)        # :51
    to_rot = rotmat_to_quat if rotation_dim == 4 else rotmat_to_rot6d                        # :60-65
    rot, trans = to_rot(TCO[:, :3, :3]), TCO[:, :3, 3].contiguous()                          # :53-54
    kp3d, kp2d = dev(other["keypoints_3d"]), dev(other["keypoints_2d"])
    if reference_keypoint_id == 0:                                                         # :76-78
        root_trans, root_rot = trans, rot
    else:                                                                                  # :80-82
        assert reference_keypoint_id < len(robot.link_names), reference_keypoint_id
        root_trans = kp3d[:, reference_keypoint_id, :]
        root_rot = robot.get_rotation_at_specific_root(pose, rot, trans, root=reference_keypoint_id)
    if use_extended_bbox:                                                                  # :43-48, 89-94
        bboxes, Kk = dev(root["bbox_gt2d_extended"]), out["root_K"]
    elif use_origin_bbox:
        bboxes, Kk = dev(input_batch["bbox_strict_bounded_original"]), dev(input_batch["K_original"])
    else:
        bboxes, Kk = dev(root["bbox_strict_bounded"]), out["root_K"]
    out["k_values"] = compute_k_values(Kk[:, 0, 0], Kk[:, 1, 1], bboxes)
    out["gt"] = dict(pose=pose, rot=rot, trans=trans, root_rot=root_rot, root_trans=root_trans,
                     root_depth=root_trans[:, 2:3], root_uv=kp2d[:, reference_keypoint_id, 0:2],
                     kp3d=kp3d, kp2d=kp2d, mask=dev(other["valid_mask_crop"]))
    return out


TERM_NAMES = ("loss_joint", "loss_rot", "loss_uv", "loss_depth", "loss_trans", "loss_error3d", "loss_error2d",
              "loss_error2d_int", "loss_error3d_int", "loss_error3d_align")          # order of function.py:313-319
_WEIGHT_KEYS = ("pose", "rot", "uv", "depth", "trans", "kp2d", "kp3d", "kp2d_int", "kp3d_int", "align_3d")


class _FusedPoseLoss(torch.autograd.Function):
    """hrp_pose_loss: the ten terms, their weighted sum and its gradient with respect to the predictions in one launch
    (the reference builds them from ~60 tensor expressions and two per-sample projection loops)."""

    @staticmethod
    def forward(ctx, K, root, image_size, wvec, gtt, pose, rot, trans, root_uv, depth, xyz_int, xyz_fk):
        import ctypes as C
        from hrpe_amd import _native as nv
        preds = [t.contiguous().float() for t in (pose, rot, trans, root_uv, depth, xyz_int, xyz_fk)]
        need = any(t.requires_grad for t in (pose, rot, trans, root_uv, depth, xyz_int, xyz_fk))
        # the seven gradients are views of ONE buffer: the backward scales it by the incoming gradient in one launch
        flat = torch.empty(sum(t.numel() for t in preds), dtype=torch.float32, device=pose.device) if need else None
        grads, off = [], 0
        for t in preds:
            grads.append(flat[off:off + t.numel()].view(t.shape) if need else None)
            off += t.numel()
        out = torch.empty(11, dtype=torch.float32, device=pose.device)
        d = nv.PoseLossDesc()
        for name, t in zip(("pose", "rot", "trans", "root_uv", "depth", "xyz_int", "xyz_fk"), preds):
            setattr(d, name, t.data_ptr())
        for name, t in zip(("gt_pose", "gt_root_rot", "gt_root_trans", "gt_root_uv", "gt_kp3d", "gt_kp2d", "mask"), gtt):
            setattr(d, name, t.data_ptr())
        d.K = K.data_ptr()
        if need:
            for name, t in zip(("d_pose", "d_rot", "d_trans", "d_root_uv", "d_depth", "d_xyz_int", "d_xyz_fk"), grads):
                setattr(d, name, t.data_ptr())
        d.out = out.data_ptr()
        for i, w in enumerate(wvec):
            d.weights[i] = w
        d.B, d.P, d.J, d.root, d.image_size = pose.shape[0], pose.shape[1], xyz_fk.shape[1], root, image_size
        d.rot_dim = rot.shape[1]
        nv.call("hrp_pose_loss", C.byref(d), torch.cuda.current_stream(pose.device).cuda_stream)
        ctx.flat, ctx.shapes = flat, [t.shape for t in preds]
        ctx.mark_non_differentiable(out)
        return out[10], out

    @staticmethod
    def backward(ctx, g_loss, _g_terms):
        if ctx.flat is None:
            return (None,) * 12
        fs, gs, off = ctx.flat * g_loss, [], 0
        for shape in ctx.shapes:
            gs.append(fs[off:off + shape.numel()].view(shape))
            off += shape.numel()
        return (None, None, None, None, None) + tuple(gs)


class _L1Loss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, gt, scale):
        from hrpe_amd import _native as nv
        p, g = pred.contiguous().float(), gt.contiguous().float()
        out = torch.empty((), dtype=torch.float32, device=pred.device)
        grad = torch.empty_like(p) if pred.requires_grad else None
        nv.call("hrp_l1_loss", p.data_ptr(), g.data_ptr(), float(scale), p.numel(), out.data_ptr(),
                grad.data_ptr() if grad is not None else None, torch.cuda.current_stream(pred.device).cuda_stream)
        ctx.grad = grad
        return out

    @staticmethod
    def backward(ctx, g_loss):
        return (ctx.grad * g_loss).view_as(ctx.grad) if ctx.grad is not None else None, None, None


SIM2REAL_YAML_WEIGHTS = dict(mask=0.0, iou=1.0, scale=0.0, align=1.0)       # configs/panda/self_supervised/*.yaml:109-112
_MASK_LOSS = {"mse_mean": 0, "bce": 1, "mse_sum": 2}


class _Sim2RealLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rendered, seg, kp3d, kp3d_int, mask_loss, w):
        import ctypes as C
        from hrpe_amd import _native as nv
        r, s_ = rendered.contiguous().float(), seg.contiguous().float()
        a, b = kp3d.contiguous().float(), kp3d_int.contiguous().float()
        B = r.shape[0]
        d = nv.Sim2RealLossDesc()
        terms = torch.empty(5, dtype=torch.float32, device=r.device)
        ws = torch.empty(8 * B, dtype=torch.float32, device=r.device)
        grads = [torch.empty_like(t) if req else None for t, req in ((r, rendered.requires_grad), (a, kp3d.requires_grad), (b, kp3d_int.requires_grad))]
        d.rendered, d.seg, d.kp3d, d.kp3d_int = r.data_ptr(), s_.data_ptr(), a.data_ptr(), b.data_ptr()
        d.B, d.HW, d.K, d.mask_loss = B, r.numel() // B, a.shape[1], mask_loss
        d.w_mask, d.w_iou, d.w_scale, d.w_align = w
        d.terms, d.workspace = terms.data_ptr(), ws.data_ptr()
        d.d_rendered, d.d_kp3d, d.d_kp3d_int = [None if g is None else g.data_ptr() for g in grads]
        nv.call("hrp_sim2real_loss", C.byref(d), torch.cuda.current_stream(r.device).cuda_stream)
        ctx.grads = [None if g is None else g.view(t.shape) for g, t in zip(grads, (rendered, kp3d, kp3d_int))]
        ctx.mark_non_differentiable(terms)
        return terms[0], terms

    @staticmethod
    def backward(ctx, g_loss, _g_terms):
        gr, ga, gb = [None if g is None else g * g_loss for g in ctx.grads]
        return gr, None, ga, gb, None, None


def sim2real_mask_loss(rendered_masks, seg_masks, pred_keypoints3d, pred_keypoints3d_int, mask_loss_func="mse_mean",
                       weights=SIM2REAL_YAML_WEIGHTS):
    """The render-and-compare losses of the self-supervised trainer (reference scripts/train_sim2real.py:435-468, BASELINE config
    5): mask (mse_mean | bce | mse_sum), IoU, scale and 3-D alignment, weighted.  rendered_masks [B, H, W] are the soft
    silhouettes of the posed robot mesh, seg_masks [B, H, W] (or [B, 1, H, W]) the segmentation network's output (detached there).
    -> (loss, dict(loss_mask, loss_iou, loss_scale, loss_error3d_align)).  Device tensors: hrp_sim2real_loss (one call, analytic
    gradient); host tensors: the same arithmetic as tensor expressions."""
    seg = seg_masks.reshape(rendered_masks.shape).detach()
    if rendered_masks.is_cuda:
        w = (float(weights["mask"]), float(weights["iou"]), float(weights["scale"]), float(weights["align"]))
        loss, t = _Sim2RealLoss.apply(rendered_masks, seg, pred_keypoints3d, pred_keypoints3d_int, _MASK_LOSS[mask_loss_func], w)
        return loss, dict(loss_mask=t[1], loss_iou=t[2], loss_scale=t[3], loss_error3d_align=t[4])
    r = rendered_masks
    if mask_loss_func == "mse_mean":
        l_mask = torch.nn.functional.mse_loss(r, seg)
    elif mask_loss_func == "bce":
        l_mask = torch.nn.functional.binary_cross_entropy(r, seg)
    else:
        l_mask = 0.001 * torch.nn.functional.mse_loss(r, seg, reduction="sum")
    inter = torch.sum(seg * r, dim=(1, 2))
    seg_area, render_area = torch.sum(seg, dim=(1, 2)), torch.sum(r, dim=(1, 2))
    l_iou = 1 - torch.mean(inter / (seg_area + render_area - inter))
    ratio = (seg_area - inter) / (render_area - inter)
    flt = (ratio.detach() > 5.0) | (ratio.detach() < 0.2)
    l_scale = torch.sum(torch.abs(torch.log(ratio)) * flt) / (torch.sum(flt) + 1e-9)
    l_align = torch.mean(torch.norm(pred_keypoints3d - pred_keypoints3d_int, dim=2))
    loss = weights["mask"] * l_mask + weights["iou"] * l_iou + weights["scale"] * l_scale + weights["align"] * l_align
    return loss, dict(loss_mask=l_mask, loss_iou=l_iou, loss_scale=l_scale, loss_error3d_align=l_align)


def depth_l1_loss(pred_depth_mm, gt_depth_m):
    """nn.L1Loss()(model(images, k_values) / 1000, gt_root_depth) of the DepthNet trainer (reference
    scripts/train_depthnet.py:231-250) as one launch with its analytic gradient (device tensors; host tensors: torch)."""
    if not pred_depth_mm.is_cuda:
        return torch.nn.functional.l1_loss(pred_depth_mm / 1000.0, gt_depth_m)
    assert pred_depth_mm.shape == gt_depth_m.shape
    return _L1Loss.apply(pred_depth_mm, gt_depth_m, 1e-3)


def full_loss(pred, gt, K, root=3, image_size=256.0, weights=FULL_YAML_WEIGHTS, kps_need_depth=None):
    """pred: the model's 8-tuple.  gt: dict(pose, root_rot, root_trans, root_uv, kp3d, kp2d, mask).
    Returns (loss, dict of the ten terms named as in function.py:313-319).  Device tensors: one fused launch
    (hrp_pose_loss, analytic gradient); host tensors: the tensor-expression form below (tests of the harness).
    multi_kp (a 9-tuple with pred_depths after pred_depth, function.py:115-117): pass kps_need_depth; the L1 term
    over the listed key-points' depths is added with weight 1 (function.py:300-311) and is not in the dict, as there."""
    if len(pred) == 9:
        assert kps_need_depth is not None, "a multi_kp prediction needs kps_need_depth"
        depths = pred[5]
        loss, terms = full_loss(pred[:5] + pred[6:], gt, K, root, image_size, weights)
        gt_depths = gt["kp3d"][:, list(kps_need_depth), 2]
        assert gt_depths.shape == depths.shape, (gt_depths.shape, depths.shape)
        return loss + torch.nn.functional.l1_loss(depths, gt_depths), terms
    if pred[0].is_cuda:
        pose, rot, trans, root_uv, depth, uvd, xyz_int, xyz_fk = pred
        f32 = lambda t: t.contiguous().float()   # noqa: E731
        gtt = [f32(gt[k]) for k in ("pose", "root_rot", "root_trans", "root_uv", "kp3d", "kp2d", "mask")]
        wvec = [float(weights[k]) for k in _WEIGHT_KEYS]
        loss, out = _FusedPoseLoss.apply(f32(K).reshape(-1, 9), int(root), float(image_size), wvec, gtt, pose, rot, trans,
                                         root_uv, depth, xyz_int, xyz_fk)
        return loss, {n: out[i] for i, n in enumerate(TERM_NAMES)}
    return full_loss_expr(pred, gt, K, root, image_size, weights)


def full_loss_expr(pred, gt, K, root=3, image_size=256.0, weights=FULL_YAML_WEIGHTS):
    """The same loss as tensor expressions (autograd) - the line-by-line restatement of function.py:191-322 the fused
    kernel is tested against."""
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
