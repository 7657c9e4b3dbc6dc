"""Oracle: forward kinematics, rot6d and camera projection in plain torch (TEST INFRASTRUCTURE).

Restates, for a kinematic tree read from a URDF file:
  * origin / rpy parsing            - reference lib/utils/urdfpytorch/utils.py:22-52, 142-167
  * joint transform J(q)            - lib/utils/urdfpytorch/urdf.py:2344-2396, 2427-2462
  * link poses, base->leaf          - lib/utils/urdfpytorch/urdf.py:3115-3140
  * actuated-joint column order     - lib/utils/urdfpytorch/urdf.py:3795-3813, 3933-3934
  * rot6d <-> R                     - lib/utils/geometries.py:100-132
  * camera-frame keypoints          - lib/utils/urdf_robot.py:82-111, 169-199
  * root rotation                   - lib/utils/urdf_robot.py:113-138
  * pinhole projection              - lib/utils/transforms.py:7-21
"""
import xml.etree.ElementTree as ET

import numpy as np
import torch

# reference lib/dataset/const.py:58-91 (data: link / joint names the keypoints are attached to)
LINK_NAMES = {
    "panda": ["panda_link0", "panda_link2", "panda_link3", "panda_link4",
              "panda_link6", "panda_link7", "panda_hand"],
    "kuka": ["iiwa_link_0", "iiwa_link_1", "iiwa_link_2", "iiwa_link_3",
             "iiwa_link_4", "iiwa_link_5", "iiwa_link_6", "iiwa_link_7"],
}
_ARM = ["s0", "s1", "e0", "e1", "w0", "w1", "w2"]
JOINT_NAMES = {
    "panda": ["panda_joint1", "panda_joint2", "panda_joint3", "panda_joint4",
              "panda_joint5", "panda_joint6", "panda_joint7", "panda_finger_joint1"],
    "kuka": ["iiwa_joint_1", "iiwa_joint_2", "iiwa_joint_3", "iiwa_joint_4",
             "iiwa_joint_5", "iiwa_joint_6", "iiwa_joint_7"],
    "baxter": ["head_pan"] + [side + "_" + j for j in _ARM for side in ("right", "left")],
}
# urdf_robot.py:61-65: baxter keypoints are the origins of these joints in their parent links
BAXTER_KEYPOINT_JOINTS = ["torso_t0"] + [side + "_" + j for j in _ARM + ["hand"] for side in ("right", "left")]


def _rpy_matrix(rpy):
    r, p, y = [float(v) for v in rpy]
    cr, cp, cy = np.cos([r, p, y])
    sr, sp, sy = np.sin([r, p, y])
    # Rz(yaw) Ry(pitch) Rx(roll), utils.py:43-52
    return np.array([[cy * cp, cy * sp * sr - cr * sy, sy * sr + cy * cr * sp],
                     [cp * sy, cy * cr + sy * sp * sr, cr * sy * sp - cy * sr],
                     [-sp, cp * sr, cp * cr]], dtype=np.float64)


class Tree:
    """Kinematic tree: joints keyed by child link, fp64 constants as parsed."""

    def __init__(self, urdf_path):
        root = ET.parse(urdf_path).getroot()
        self.links = [n.attrib["name"] for n in root.findall("link")]
        self.joint_of_child = {}
        joints = []
        for n in root.findall("joint"):
            origin = np.eye(4, dtype=np.float64)
            o = n.find("origin")
            if o is not None:
                if "xyz" in o.attrib:
                    origin[:3, 3] = np.array(o.attrib["xyz"].split(), dtype=np.float64)
                if "rpy" in o.attrib:
                    origin[:3, :3] = _rpy_matrix(o.attrib["rpy"].split())
            ax = n.find("axis")
            axis = np.array(ax.attrib["xyz"].split(), dtype=np.float64) if ax is not None else None
            mim = n.find("mimic")
            j = dict(name=n.attrib["name"], type=n.attrib["type"],
                     parent=n.find("parent").attrib["link"], child=n.find("child").attrib["link"],
                     origin=origin, axis=axis,
                     mimic=None if mim is None else (mim.attrib["joint"],
                                                     float(mim.attrib.get("multiplier", 1.0)),
                                                     float(mim.attrib.get("offset", 0.0))))
            joints.append(j)
            self.joint_of_child[j["child"]] = j
        self.joints = joints
        children = set(self.joint_of_child)
        self.base = [l for l in self.links if l not in children][0]
        # actuated joints sorted by distance of the child link from the base (urdf.py:3795-3813)
        act = [j for j in joints if j["type"] != "fixed" and j["mimic"] is None]
        depth = [len(self.path_to_base(j["child"])) for j in act]
        self.actuated = [act[i] for i in np.argsort(depth, kind="stable")]
        self.cfg_index = {j["name"]: i for i, j in enumerate(self.actuated)}

    def path_to_base(self, link):
        path = [link]
        while path[-1] != self.base:
            path.append(self.joint_of_child[path[-1]]["parent"])
        return path


def joint_pose(j, q):
    """J(q) [B,4,4] fp32 for one joint; q is [B] or None (urdf.py:2344-2396, 2427-2462)."""
    origin = torch.as_tensor(j["origin"]).to(torch.float32)
    if j["type"] == "fixed" or q is None:
        return origin
    B = q.shape[0]
    M = torch.eye(4, dtype=torch.float32).repeat(B, 1, 1)
    if j["type"] in ("revolute", "continuous"):
        a = j["axis"] / np.linalg.norm(j["axis"])
        s, c = torch.sin(q), torch.cos(q)
        outer = torch.as_tensor(np.outer(a, a)).to(torch.float32)
        skew = torch.as_tensor(np.array([[0.0, -a[2], a[1]], [a[2], 0.0, -a[0]],
                                         [-a[1], a[0], 0.0]])).to(torch.float32)
        R = c[:, None, None] * torch.eye(3) + (1.0 - c)[:, None, None] * outer \
            + s[:, None, None] * skew
        M[:, :3, :3] = R
    elif j["type"] == "prismatic":
        M[:, :3, 3] = torch.as_tensor(j["axis"]).to(torch.float32) * q[:, None]
    else:
        raise NotImplementedError(j["type"])
    return origin @ M


def link_poses(tree, q, links):
    """T_base->link [B,4,4] for each requested link (urdf.py:3115-3140)."""
    B = q.shape[0]
    cache = {tree.base: torch.eye(4, dtype=torch.float32).repeat(B, 1, 1)}

    def pose(link):
        if link in cache:
            return cache[link]
        j = tree.joint_of_child[link]
        if j["mimic"] is not None:
            src, mul, off = j["mimic"]
            qv = mul * q[:, tree.cfg_index[src]] + off
        elif j["name"] in tree.cfg_index:
            qv = q[:, tree.cfg_index[j["name"]]]
        else:
            qv = None
        cache[link] = pose(j["parent"]) @ joint_pose(j, qv)
        return cache[link]

    return torch.stack([pose(l) for l in links], dim=1)


def rot6d_to_rotmat(r):
    """geometries.py:100-115: rows x, y, z with x = a/|a|, z = (x x b)/|.|, y = z x x."""
    a, b = r[..., 0:3], r[..., 3:6]
    x = a / torch.norm(a, dim=-1, keepdim=True)
    z = torch.cross(x, b, dim=-1)
    z = z / torch.norm(z, dim=-1, keepdim=True)
    y = torch.cross(z, x, dim=-1)
    return torch.stack((x, y, z), dim=-2)


def rotmat_to_rot6d(R):
    """geometries.py:117-132: first two rows."""
    return R[..., :2, :].reshape(*R.shape[:-2], 6)


def quat_to_rotmat(quat):
    """geometries.py:21-41: (w, x, y, z), normalised by (norm + 1e-9)."""
    nq = quat / (quat.norm(p=2, dim=1, keepdim=True) + 1e-9)
    w, x, y, z = nq[:, 0], nq[:, 1], nq[:, 2], nq[:, 3]
    w2, x2, y2, z2 = w * w, x * x, y * y, z * z
    wx, wy, wz, xy, xz, yz = w * x, w * y, w * z, x * y, x * z, y * z
    return torch.stack([w2 + x2 - y2 - z2, 2 * xy - 2 * wz, 2 * wy + 2 * xz,
                        2 * wz + 2 * xy, w2 - x2 + y2 - z2, 2 * yz - 2 * wx,
                        2 * xz - 2 * wy, 2 * wx + 2 * yz, w2 - x2 - y2 + z2], dim=1).view(-1, 3, 3)


def rotmat_to_quat(m):
    """geometries.py:63-82: w = sqrt(max(1 + trace, 0)) / 2 clamped at 1e-8, x y z from the antisymmetric part, normalised
    (norm clamped at 1e-8)."""
    w = torch.sqrt(torch.clamp(1.0 + m[:, 0, 0] + m[:, 1, 1] + m[:, 2, 2], min=0.0)) / 2.0
    w = torch.clamp(w, min=1e-8)
    w4 = 4.0 * w
    q = torch.stack([w, (m[:, 2, 1] - m[:, 1, 2]) / w4, (m[:, 0, 2] - m[:, 2, 0]) / w4, (m[:, 1, 0] - m[:, 0, 1]) / w4], 1)
    return q / torch.clamp(torch.sqrt((q * q).sum(1, keepdim=True)), min=1e-8)


def _base2cam(rot6d, trans):
    """rot6d: [B, 6] (Zhou et al.) or [B, 4] quaternion (urdf_robot.py:86-92)."""
    B = rot6d.shape[0]
    T = torch.zeros(B, 4, 4, dtype=torch.float32)
    T[:, :3, :3] = rot6d_to_rotmat(rot6d) if rot6d.shape[1] == 6 else quat_to_rotmat(rot6d)
    T[:, :3, 3] = trans
    T[:, 3, 3] = 1.0
    return T


class Robot:
    """URDFRobot restated: keypoints of a robot in the camera frame (urdf_robot.py:22-199)."""

    def __init__(self, urdf_path, robot_type="panda"):
        self.tree = Tree(urdf_path)
        self.dof = len(JOINT_NAMES[robot_type])
        assert [j["name"] for j in self.tree.actuated] == JOINT_NAMES[robot_type]
        if robot_type == "baxter":  # urdf_robot.py:57-74
            by_name = {j["name"]: j for j in self.tree.joints}
            self.link_names = [by_name[n]["parent"] for n in BAXTER_KEYPOINT_JOINTS]
            self.offsets = torch.as_tensor(np.stack([by_name[n]["origin"][:3, 3] for n in BAXTER_KEYPOINT_JOINTS])).float()
        else:  # panda / kuka: urdf_robot.py:53-56
            self.link_names = LINK_NAMES[robot_type]
            self.offsets = torch.zeros(len(self.link_names), 3)

    def get_TWL(self, q):
        return link_poses(self.tree, q, self.link_names)

    def _points(self, TWL):
        return (TWL[:, :, :3, :3] @ self.offsets[None, :, :, None]).squeeze(-1) + TWL[:, :, :3, 3]

    def get_keypoints_only_fk(self, q):
        return self._points(self.get_TWL(q))

    def get_keypoints(self, q, rot6d, trans):
        return self._points(_base2cam(rot6d, trans)[:, None] @ self.get_TWL(q))

    def get_keypoints_root(self, q, rot6d, trans, root=0):
        if root == 0:
            return self.get_keypoints(q, rot6d, trans)
        TWL = self.get_TWL(q)
        TWL = torch.linalg.inv(TWL[:, root:root + 1]) @ TWL      # urdf_robot.py:194-196
        return self._points(_base2cam(rot6d, trans)[:, None] @ TWL)

    def get_rotation_at_specific_root(self, q, rot6d, trans, root=0):
        if root == 0:
            return rot6d
        TWL = _base2cam(rot6d, trans)[:, None] @ self.get_TWL(q)
        if rot6d.shape[1] == 4:            # urdf_robot.py:136-137
            return rotmat_to_quat(TWL[:, root, :3, :3])
        return rotmat_to_rot6d(TWL[:, root, :3, :3])


MESH_LINKS = {"panda": ["panda_link%d" % i for i in range(8)] + ["panda_hand"]}      # urdf_robot.py:209-219


def pose_mesh(robot, q, rot6d, trans, verts, vert_link, root=0, robot_type="panda"):
    """Camera-frame vertices of the posed mesh [B, V, 3].  mesh_renderer.py:126-173 (every link's vertices `verts @ R.T + t`
    with the link's pose) viewed by the camera of urdf_robot.py:242-275: base-to-camera pose, re-rooted at key-point `root`
    (`base2cam @ inv(TWL_root)`, :266-271), mirrored through the origin when its translation has negative depth (:250-253;
    pytorch3d's `X @ R + T` with R transposed at :245 is the column form `R X + T` used here)."""
    TL = link_poses(robot.tree, q, MESH_LINKS[robot_type])                        # [B, L, 4, 4]
    M = _base2cam(rot6d, trans)
    if root != 0:
        M = M @ torch.linalg.inv(robot.get_TWL(q)[:, root])
    flip = torch.where(M[:, 2, 3] < 0, -1.0, 1.0)[:, None, None]
    R, T = M[:, :3, :3] * flip, M[:, :3, 3] * flip[:, :, 0]
    vl = vert_link.long()
    posed = torch.einsum("bvkj,vj->bvk", TL[:, vl, :3, :3], verts) + TL[:, vl, :3, 3]
    return torch.einsum("bkj,bvj->bvk", R, posed) + T[:, None, :]


def project(K, pts):
    """point_projection_from_3d_tensor, transforms.py:7-21: uv = (K p)[:2] / (K p)[2]."""
    h = torch.einsum("bij,bkj->bki", K, pts)
    return h[..., :2] / h[..., 2:3]
