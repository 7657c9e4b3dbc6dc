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
