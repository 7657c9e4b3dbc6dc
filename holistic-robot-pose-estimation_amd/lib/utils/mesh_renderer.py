"""Renderer of the self-supervised trainer with the reference's names (lib/utils/mesh_renderer.py:61-173, lib/utils/urdf_robot.py:201-275).

The reference builds a pytorch3d MeshRenderer per call and renders one sample at a time (scripts/train_sim2real.py:405-418: the
mesh of every sample is posed on the CPU, then rasterised on the GPU).  Here the renderer object only carries the camera and the
mesh - vertices in their links' frames, the link of every vertex, faces - and the whole batch is posed and rasterised by
URDFRobot.render_silhouette (hrp_mesh_pose + hrp_silhouette_fwd).  PARITY UNPINNED for the rasteriser (csrc/silhouette.hip)."""
import os

import numpy as np
import torch


def load_mesh_files(mesh_files):
    """Wavefront .obj files, one per visual-mesh link (reference mesh_renderer.py:78-84 preloads them with pytorch3d's load_obj) ->
    (verts [V, 3] float32 in link frames, vert_link [V] uint8, faces [F, 3] int32); polygons are fanned into triangles."""
    verts, links, faces = [], [], []
    for li, path in enumerate(mesh_files):
        if not os.path.exists(path):
            raise FileNotFoundError(f"mesh file {path} (the reference's visual meshes are not part of this repository)")
        base = len(verts)
        with open(path) as fh:
            for line in fh:
                if line.startswith("v "):
                    verts.append([float(v) for v in line.split()[1:4]])
                    links.append(li)
                elif line.startswith("f "):
                    idx = [int(tok.split("/")[0]) for tok in line.split()[1:]]
                    idx = [base + i - 1 if i > 0 else len(verts) + i for i in idx]
                    for k in range(1, len(idx) - 1):
                        faces.append([idx[0], idx[k], idx[k + 1]])
    return (torch.tensor(np.asarray(verts, np.float32).reshape(-1, 3)), torch.tensor(np.asarray(links, np.uint8)),
            torch.tensor(np.asarray(faces, np.int32).reshape(-1, 3)))


class RobotMeshRenderer:
    """focal_length = [-fx, -fy] and principal_point = [cx, cy] of the RENDERED image, image_size = (H, W) - the arguments
    URDFRobot.set_robot_renderer passes (urdf_robot.py:221-227: the negative focal lengths undo pytorch3d's left / up axes, so
    the image is the ordinary pinhole one).  mesh: (verts, vert_link, faces) or None to load `mesh_files`."""

    def __init__(self, focal_length, principal_point, image_size, robot=None, mesh_files=None, device="cuda", mesh=None,
                 sigma=1e-8, blur_radius=None):
        self.image_size = (int(image_size[0]), int(image_size[1]))
        self.device = torch.device(device)
        fx, fy = abs(float(focal_length[0])), abs(float(focal_length[1]))
        self.K = torch.tensor([[fx, 0.0, float(principal_point[0])], [0.0, fy, float(principal_point[1])], [0.0, 0.0, 1.0]],
                              device=self.device)
        self.robot = robot
        self.sigma = sigma                                                    # BlendParams(sigma = 1e-8), mesh_renderer.py:94
        self.blur_radius = float(np.log(1.0 / 1e-4 - 1.0) * sigma) if blur_radius is None else blur_radius     # :97
        m = mesh if mesh is not None else load_mesh_files(mesh_files)
        self.mesh = tuple(t.to(self.device) for t in m)

    def silhouettes(self, robot, joint_angles, rot, trans, root=0):
        """[B, H, W] soft silhouettes (channel 3 of the reference's silhouette_renderer output, urdf_robot.py:257)."""
        B = joint_angles.shape[0]
        return robot.render_silhouette(joint_angles, rot, trans, self.mesh, self.K.expand(B, 3, 3), self.image_size, root=root,
                                       sigma=self.sigma, blur_radius=self.blur_radius)
