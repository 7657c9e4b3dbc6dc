"""Validation metrics on the device (drop-in for reference lib/utils/metrics.py:8-162).

The reference moves every prediction to the host (`.detach().cpu().numpy()`, metrics.py:31-43) and evaluates the
AUC curves with a 10 000-iteration Python loop (:125-146).  Here ``compute_metrics_batch`` keeps everything on the
GPU (FK and projection through the HIP kernels, the rest O(B * keypoints) tensor expressions) and returns device
tensors - the caller decides when to synchronise - and ``summary_add_pck`` evaluates the same step curves with one
sort + searchsorted.  Same names, arguments and return order as the reference.
"""
import torch

from hrpe_amd.lib.utils.transforms import point_projection_from_3d_tensor

ADD_THRESHOLDS_MM = [1, 5, 10, 20, 40, 60, 80, 100]                    # metrics.py:50, 120
PCK_THRESHOLDS_PX = [2.5, 5.0, 7.5, 10.0, 12.5, 15.0, 17.5, 20.0]      # metrics.py:51, 121


def compute_metrics_batch(robot, gt_keypoints3d, gt_keypoints2d, K_original, gt_joint, **pred_kwargs):
    """Returns (error3d [B], error2d [B], dis3d [nkp], dis2d [nkp], l1_jointerror [dof], mean_jointerror [B],
    error_depth [B], batch_error_relative [B], error3d_relative [B]) as fp32 device tensors (metrics.py:8-113)."""
    pred_joint, pred_rot, pred_trans = pred_kwargs["pred_joint"], pred_kwargs["pred_rot"], pred_kwargs["pred_trans"]
    if pred_kwargs.get("pred_xy") is not None and pred_kwargs.get("pred_depth") is not None:          # :15-18
        pred_trans = torch.cat((pred_kwargs["pred_xy"], pred_kwargs["pred_depth"]), dim=-1)
    pred_xyz_integral = pred_kwargs["pred_xyz_integral"]
    root = pred_kwargs["reference_keypoint_id"]
    with torch.no_grad():
        if pred_joint is None or pred_rot is None or pred_trans is None:                               # :22-25
            assert pred_xyz_integral is not None
            pred3d = pred_xyz_integral.detach().float()
            pred_joint = None
        elif root == 0:                                                                                # :27-30
            pred3d = robot.get_keypoints(pred_joint.detach(), pred_rot.detach(), pred_trans.detach())
        else:                                                                                          # :32-34
            pred3d = robot.get_keypoints_root(pred_joint.detach(), pred_rot.detach(), pred_trans.detach(), root=root)
        B, nkp = pred3d.shape[0], len(robot.link_names)
        gt3d, gt2d = gt_keypoints3d.detach().float(), gt_keypoints2d.detach().float()
        pred2d = point_projection_from_3d_tensor(K_original.detach().float(), pred3d)                  # :42
        assert pred3d.shape == (B, nkp, 3) and gt3d.shape == (B, nkp, 3), (pred3d.shape, gt3d.shape)
        assert pred2d.shape == (B, nkp, 2) and gt2d.shape == (B, nkp, 2), (pred2d.shape, gt2d.shape)
        error3d_batch = torch.norm(pred3d - gt3d, dim=2)                                               # :54-56
        error3d = error3d_batch.mean(dim=1)
        error2d_batch = torch.norm(pred2d - gt2d, dim=2)                                               # :60
        valid = (gt2d[:, :, 0] <= 640.0) & (gt2d[:, :, 0] >= 0) & (gt2d[:, :, 1] <= 480.0) & (gt2d[:, :, 1] >= 0)
        error2d_all = error2d_batch * valid
        error2d = error2d_all.sum(dim=1) / valid.sum(dim=1)                                            # :63-66
        dis3d = error3d_batch.mean(dim=0)                                                              # :70
        dis2d = error2d_all.sum(dim=0) / valid.sum(dim=0)                                              # :71-73
        if pred_joint is not None:                                                                     # :78-88
            gj, pj = gt_joint.detach().float(), pred_joint.detach().float()
            assert gj.shape == pj.shape == (B, robot.dof), (pj.shape, gj.shape)
            ej = (gj - pj).abs()
            l1_jointerror = ej.mean(dim=0)
            mean_jointerror = (ej[:, :-1] if robot.robot_type == "panda" else ej).mean(dim=1)
        else:                                                                                          # :89-91
            l1_jointerror = torch.zeros(robot.dof, device=pred3d.device)
            mean_jointerror = torch.zeros(B, device=pred3d.device)
        error_depth = (pred3d[:, root, 2] - gt3d[:, root, 2]).abs()                                    # :95
        pred_rel = pred3d[:, :, 2] - pred3d[:, root:root + 1, 2]                                       # :98-101
        gt_rel = gt3d[:, :, 2] - gt3d[:, root:root + 1, 2]
        batch_error_relative = (pred_rel - gt_rel).abs().mean(dim=1)
        d = pred3d - gt3d                                                                              # :104-110
        d = torch.cat((d[:, :, :2], (pred_rel - gt_rel).unsqueeze(-1)), dim=2)
        error3d_relative = torch.norm(d, dim=2).mean(dim=1)
    return (error3d, error2d, dis3d, dis2d, l1_jointerror, mean_jointerror, error_depth, batch_error_relative,
            error3d_relative)


def _auc(dis_sorted, limit, delta):
    """mean_i trapz of the step curve c(v) = mean(dis <= v) sampled at v = k * delta, k = 0 .. limit/delta - 1
    (metrics.py:125-146); the comparison runs in the array's fp32 like numpy's does with a Python scalar."""
    n = int(round(limit / delta))
    v = torch.arange(n, dtype=torch.float64, device=dis_sorted.device) * delta
    counts = torch.searchsorted(dis_sorted, v.to(dis_sorted.dtype), right=True).double() / dis_sorted.numel()
    return ((counts.sum() - 0.5 * (counts[0] + counts[-1])) * delta / limit).item()


def summary_add_pck(alldis):
    """alldis: {'dis3d': ..., 'dis2d': ...} (lists of per-image values or tensors) -> the reference's summary dict."""
    def flat(x):
        if isinstance(x, torch.Tensor):
            return x.detach().float().reshape(-1)
        if len(x) and isinstance(x[0], torch.Tensor):
            return torch.cat([t.detach().float().reshape(-1) for t in x])
        return torch.as_tensor(x, dtype=torch.float32).reshape(-1)
    dis3d, dis2d = flat(alldis["dis3d"]), flat(alldis["dis2d"])
    assert dis3d.shape[0] == dis2d.shape[0]
    s3, s2 = torch.sort(dis3d).values, torch.sort(dis2d).values
    summary = {
        "ADD/mean": dis3d.mean().item(), "ADD/median": _median(s3), "ADD/AUC": _auc(s3, 0.1, 0.00001),
        "ADD_2D/mean": dis2d.mean().item(), "ADD_2D/median": _median(s2), "PCK/AUC": _auc(s2, 20.0, 0.01),
    }
    for th in ADD_THRESHOLDS_MM:
        summary[f"ADD_{th}_mm"] = (dis3d <= th * 1e-3).float().mean().item()
    for th in PCK_THRESHOLDS_PX:
        summary[f"PCK_{th}_pixel"] = (dis2d <= th).float().mean().item()
    return summary


def _median(s):
    """np.median of a sorted vector (mean of the two middle elements for even length)."""
    n = s.numel()
    return (s[n // 2].item() if n % 2 else 0.5 * (s[n // 2 - 1].item() + s[n // 2].item()))
