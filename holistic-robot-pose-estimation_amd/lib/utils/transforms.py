"""Camera helpers with the reference's names (lib/utils/transforms.py).

``point_projection_from_3d_tensor`` replaces the reference's per-sample Python loop
(transforms.py:17-21) by one kernel launch (csrc/heads.hip: project_fwd/bwd)."""
import torch

from hrpe_amd import _native as nv


class _ProjectFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, K, pts):
        if pts.device.type != "cuda":
            raise nv.HrpError("point_projection_from_3d_tensor runs on the GPU only (no CPU path)")
        K = K.contiguous().float()
        pts = pts.contiguous().float()
        B, P = pts.shape[0], pts.shape[1]
        uv = torch.empty(B, P, 2, device=pts.device)
        s = torch.cuda.current_stream(pts.device).cuda_stream
        nv.call("hrp_project_fwd", K.data_ptr(), pts.data_ptr(), B, P, uv.data_ptr(), s)
        ctx.save_for_backward(K, pts)
        return uv

    @staticmethod
    def backward(ctx, g):
        K, pts = ctx.saved_tensors
        B, P = pts.shape[0], pts.shape[1]
        g = g.contiguous().float()
        d = torch.empty_like(pts)
        s = torch.cuda.current_stream(pts.device).cuda_stream
        nv.call("hrp_project_bwd", K.data_ptr(), pts.data_ptr(), g.data_ptr(), B, P, d.data_ptr(), s)
        return None, d


def point_projection_from_3d_tensor(camera_K, points):
    """camera_K [B,3,3], points [B,P,3] -> [B,P,2] = (K p)[:2] / (K p)[2]."""
    return _ProjectFn.apply(camera_K, points)
