"""Soft-silhouette rasteriser restated in torch (TEST INFRASTRUCTURE ONLY - imported by tests/, never by the product path).

PARITY UNPINNED: the reference renders with pytorch3d 0.7.4 (lib/utils/mesh_renderer.py:78-109: MeshRasterizer with
blur_radius = log(1 / 1e-4 - 1) * sigma, faces_per_pixel = 100, SoftSilhouetteShader with BlendParams(sigma = 1e-8), PerspectiveCameras
with focal (-fx, -fy), in_ndc = False); pytorch3d is neither in the reference tree nor in the build container and the reference holds
no rendered fixture.  This file restates pytorch3d's published algorithm - rasterize_meshes: a face is kept at a pixel when the pixel
centre is inside it or closer than blur_radius (squared NDC distance to the nearest edge segment), signed distance negative inside;
sigmoid_alpha_blend: alpha = 1 - prod(1 - sigmoid(-dist / sigma)) - and is what csrc/silhouette.hip is tested against.
faces_per_pixel = 100: pytorch3d keeps the 100 nearest kept faces per pixel; with at most 100 kept faces (always, for a robot mesh:
`return_counts` gives the number and the GPU tests assert it) that is every kept face, which is what both restatements use.
Near plane: the reference's settings (z_clip_value None, PerspectiveCameras without znear) switch pytorch3d's clipping OFF; its
rasteriser then drops a face when any vertex has z < 1e-8 - the `zmin >= 1e-8` rule below."""
import torch


def _seg_dist2(p, a, b):
    """squared distance from points p [P, 1, 2] to segments (a, b) [1, F, 2] -> [P, F]"""
    ab = b - a
    l2 = (ab * ab).sum(-1)
    t = ((p - a) * ab).sum(-1) / torch.where(l2 > 1e-8, l2, torch.ones_like(l2))
    t = torch.where(l2 > 1e-8, t.clamp(0.0, 1.0), torch.ones_like(t))
    q = a + t[..., None] * ab - p
    return (q * q).sum(-1)


def soft_silhouette(uv, z, faces, H, W, sigma=1e-8, blur_radius=None, return_counts=False):
    """uv [B, V, 2] pixel coordinates (u = fx X / Z + cx), z [B, V] depths, faces [F, 3] -> alpha [B, H, W] (differentiable in uv)."""
    if blur_radius is None:
        blur_radius = float(torch.log(torch.tensor(1.0 / 1e-4 - 1.0)) * sigma)
    k2 = (2.0 / min(H, W)) ** 2
    ys, xs = torch.meshgrid(torch.arange(H, dtype=uv.dtype), torch.arange(W, dtype=uv.dtype), indexing="ij")
    pc = torch.stack([xs.reshape(-1) + 0.5, ys.reshape(-1) + 0.5], -1)[:, None, :]            # [P, 1, 2]
    out, counts = [], []
    for b in range(uv.shape[0]):
        v0, v1, v2 = [uv[b, faces[:, i]][None] for i in range(3)]                             # [1, F, 2]
        zmin = z[b, faces].min(-1).values

        def edge(p, a, c):
            return (p[..., 0] - a[..., 0]) * (c[..., 1] - a[..., 1]) - (p[..., 1] - a[..., 1]) * (c[..., 0] - a[..., 0])
        area = edge(v2, v0, v1)                                                               # [1, F]
        ok = (zmin >= 1e-8) & ((area * k2).abs() > 1e-8)[0]
        safe = torch.where(area.abs() > 0, area, torch.ones_like(area))
        w0, w1, w2 = edge(pc, v1, v2) / safe, edge(pc, v2, v0) / safe, edge(pc, v0, v1) / safe
        inside = (w0 > 0) & (w1 > 0) & (w2 > 0)
        dn = torch.minimum(torch.minimum(_seg_dist2(pc, v0, v1), _seg_dist2(pc, v1, v2)), _seg_dist2(pc, v2, v0)) * k2
        kept = ok[None] & (inside | (dn < blur_radius))
        s = torch.where(inside, -dn, dn)
        logq = torch.nn.functional.logsigmoid(s / sigma).clamp(min=-100.0)                    # log(1 - sigmoid(-s / sigma))
        logp = torch.where(kept, logq, torch.zeros_like(logq)).sum(-1)
        out.append((1.0 - torch.exp(logp)).reshape(H, W))
        counts.append(kept.sum(-1).reshape(H, W))
    if return_counts:
        return torch.stack(out), torch.stack(counts)
    return torch.stack(out)
