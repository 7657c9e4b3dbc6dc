"""Rotation helpers with the reference's names (lib/utils/geometries.py:100-132).

These are caller-side conveniences on tiny [B, 6] / [B, 3, 3] tensors (ground-truth preparation in
lib/core/function.py); inside the model the 6-D -> matrix step is part of the FK kernel (csrc/heads.hip).
"""
import torch


def rot6d_to_rotmat(poses):
    """Zhou et al. 6-D representation -> rotation matrix with ROWS x, y, z (geometries.py:100-115)."""
    assert poses.shape[-1] == 6
    a, b = poses[..., 0:3], poses[..., 3:6]
    x = a / torch.norm(a, p=2, dim=-1, keepdim=True)
    z = torch.cross(x, b, dim=-1)
    z = z / torch.norm(z, p=2, dim=-1, keepdim=True)
    y = torch.cross(z, x, dim=-1)
    return torch.stack((x, y, z), dim=-2)


def rotmat_to_rot6d(matrix):
    """First two rows, flattened (geometries.py:117-132)."""
    return matrix[..., :2, :].clone().reshape(*matrix.size()[:-2], 6)


def quat_to_rotmat(quat):
    """Quaternion (w, x, y, z) -> rotation matrix, normalised by (norm + 1e-9) (geometries.py:21-41)."""
    nq = quat / (quat.norm(p=2, dim=1, keepdim=True) + 1e-9)
    w, x, y, z = nq[:, 0], nq[:, 1], nq[:, 2], nq[:, 3]
    w2, x2, y2, z2 = w * w, x * x, y * y, z * z
    wx, wy, wz, xy, xz, yz = w * x, w * y, w * z, x * y, x * z, y * z
    return torch.stack([w2 + x2 - y2 - z2, 2 * xy - 2 * wz, 2 * wy + 2 * xz,
                        2 * wz + 2 * xy, w2 - x2 + y2 - z2, 2 * yz - 2 * wx,
                        2 * xz - 2 * wy, 2 * wx + 2 * yz, w2 - x2 - y2 + z2], dim=1).view(-1, 3, 3)


def rotmat_to_quat(matrices):
    """Rotation matrix -> unit quaternion (w, x, y, z) with w >= 1e-8 (geometries.py:63-82)."""
    m = matrices
    w = torch.sqrt(torch.clamp(1.0 + m[:, 0, 0] + m[:, 1, 1] + m[:, 2, 2], min=0.0)) / 2.0
    w = torch.clamp(w, min=1e-8)
    w4 = 4.0 * w
    q = torch.stack([w, (m[:, 2, 1] - m[:, 1, 2]) / w4, (m[:, 0, 2] - m[:, 2, 0]) / w4, (m[:, 1, 0] - m[:, 0, 1]) / w4], 1)
    return q / torch.clamp(torch.sqrt((q * q).sum(1, keepdim=True)), min=1e-8)
