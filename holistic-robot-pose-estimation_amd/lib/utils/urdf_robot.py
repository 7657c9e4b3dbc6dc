"""Kinematics-only robot model on the HIP FK kernel (drop-in for the FK part of reference
lib/utils/urdf_robot.py:22-199; the rendering half of that file is out of scope).

``URDFRobot(robot_type)`` parses the URDF once on the host into a flat chain descriptor
(``hrp_fk_chain`` in include/hrp.h) and every ``get_keypoints*`` call is ONE launch of the
wavefront-per-sample FK kernel (csrc/heads.hip) instead of ~50 small torch matmuls per call
(reference lib/utils/urdfpytorch/urdf.py:3115-3140).
"""
import ctypes as C
import os
import xml.etree.ElementTree as ET

import numpy as np
import torch

from hrpe_amd import _native as nv
from hrpe_amd.lib.dataset.const import BAXTER_KEYPOINT_JOINTS, JOINT_NAMES, LINK_NAMES, MESH_LINKS

_ASSETS = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "assets")


def _default_urdf(robot_type):
    # reference location first (lib/config.py:33), then the kinematics-only file shipped with this package
    cands = {"panda": ["data/deps/panda-description/panda.urdf", os.path.join(_ASSETS, "panda_kinematics.urdf")],
             "kuka": ["data/deps/kuka-description/iiwa_description/urdf/iiwa7.urdf",
                      os.path.join(_ASSETS, "kuka_kinematics.urdf")],
             "baxter": ["data/deps/baxter-description/baxter_description/urdf/baxter.urdf",
                        os.path.join(_ASSETS, "baxter_kinematics.urdf")]}
    for c in cands.get(robot_type, []):
        if os.path.isfile(c):
            return c
    raise FileNotFoundError(f"no URDF for robot '{robot_type}' (looked in {cands.get(robot_type)})")


def _rpy(r, p, y):
    cr, cp, cy, sr, sp, sy = np.cos(r), np.cos(p), np.cos(y), np.sin(r), np.sin(p), np.sin(y)
    return np.array([[cy * cp, cy * sp * sr - cr * sy, sy * sr + cy * cr * sp],
                     [cp * sy, cy * cr + sy * sp * sr, cr * sy * sp - cy * sr],
                     [-sp, cp * sr, cp * cr]], dtype=np.float64)


def parse_chain(urdf_path, link_names, offsets=None):
    """URDF -> (ctypes hrp_fk_chain, actuated joint names in configuration-column order)."""
    root = ET.parse(urdf_path).getroot()
    joints = []
    for n in root.findall("joint"):
        T = np.eye(4)
        o = n.find("origin")
        if o is not None:
            if "xyz" in o.attrib:
                T[:3, 3] = [float(v) for v in o.attrib["xyz"].split()]
            if "rpy" in o.attrib:
                T[:3, :3] = _rpy(*[float(v) for v in o.attrib["rpy"].split()])
        ax = n.find("axis")
        mim = n.find("mimic")
        joints.append(dict(name=n.attrib["name"], type=n.attrib["type"], parent=n.find("parent").attrib["link"],
                           child=n.find("child").attrib["link"], origin=T,
                           axis=np.array([float(v) for v in ax.attrib["xyz"].split()]) if ax is not None else np.array([1.0, 0, 0]),
                           mimic=None if mim is None else (mim.attrib["joint"], float(mim.attrib.get("multiplier", 1.0)),
                                                           float(mim.attrib.get("offset", 0.0)))))
    by_child = {j["child"]: j for j in joints}

    def depth(link):
        d = 0
        while link in by_child:
            link = by_child[link]["parent"]
            d += 1
        return d

    order = sorted(range(len(joints)), key=lambda i: depth(joints[i]["child"]))  # parents before children
    joints = [joints[i] for i in order]
    index = {j["child"]: i for i, j in enumerate(joints)}
    # configuration columns: non-fixed, non-mimic joints by ascending distance from the base
    # (reference urdf.py:3795-3813, 3933-3934)
    act = [j for j in joints if j["type"] != "fixed" and j["mimic"] is None]
    cfg_of = {j["name"]: i for i, j in enumerate(act)}
    if len(joints) > nv.FK_MAX_JOINTS or len(link_names) > nv.FK_MAX_KP:
        raise ValueError("robot too large for hrp_fk_chain")
    ch = nv.FkChain()
    ch.njoints = len(joints)
    for i, j in enumerate(joints):
        ch.parent[i] = index.get(j["parent"], -1)
        ch.type[i] = {"fixed": 0, "revolute": 1, "continuous": 1, "prismatic": 2}[j["type"]]
        ch.mimic_mul[i], ch.mimic_off[i] = 1.0, 0.0
        if j["type"] == "fixed":
            ch.cfg[i] = -1
        elif j["mimic"] is not None:
            src, mul, off = j["mimic"]
            ch.cfg[i] = cfg_of.get(src, -1)
            ch.mimic_mul[i], ch.mimic_off[i] = mul, off
        else:
            ch.cfg[i] = cfg_of[j["name"]]
        for r in range(3):
            for c in range(4):
                ch.origin[i][r * 4 + c] = float(j["origin"][r, c])
        a = j["axis"] / np.linalg.norm(j["axis"])
        for r in range(3):
            ch.axis[i][r] = float(a[r])
    ch.nkp = len(link_names)
    for k, ln in enumerate(link_names):
        ch.kp_frame[k] = index.get(ln, -1)
        for r in range(3):
            ch.kp_offset[k][r] = float(offsets[k][r]) if offsets is not None else 0.0
    ch.dof = len(act)
    return ch, [j["name"] for j in act]


class _FKFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, robot, q, rot, trans, K, root, want):
        B = q.shape[0]
        dev = q.device
        chain = robot.chain_on(dev)
        q, rot, trans = [t.contiguous().float() for t in (q, rot, trans)]
        Kc = K.contiguous().float() if K is not None else None
        nkp = robot.nkp
        xyz = torch.empty(B, nkp, 3, device=dev)
        uv = torch.empty(B, nkp, 2, device=dev) if Kc is not None else None
        rd = rot.shape[1]               # 6: two rows of the rotation matrix, 4: quaternion (urdf_robot.py:86-92)
        rr = torch.empty(B, rd, device=dev)
        s = torch.cuda.current_stream(dev).cuda_stream
        nv.call("hrp_fk_project_rot_fwd", chain.data_ptr(), q.data_ptr(), rot.data_ptr(), rd, trans.data_ptr(),
                Kc.data_ptr() if Kc is not None else None, B, root, xyz.data_ptr(),
                uv.data_ptr() if uv is not None else None, rr.data_ptr(), s)
        ctx.save_for_backward(q, rot, trans, Kc if Kc is not None else torch.empty(0, device=dev))
        ctx.robot, ctx.root, ctx.hasK = robot, root, Kc is not None
        outs = {"xyz": xyz, "uv": uv, "rot": rr}
        ctx.want = want
        if "rot" in want:
            ctx.mark_non_differentiable(rr)
        return tuple(outs[w] for w in want)

    @staticmethod
    def backward(ctx, *grads):
        q, rot, trans, Kc = ctx.saved_tensors
        B, dev = q.shape[0], q.device
        g = dict(zip(ctx.want, grads))
        gx = g.get("xyz")
        gu = g.get("uv")
        gx = gx.contiguous().float() if gx is not None else None
        gu = gu.contiguous().float() if gu is not None else None
        dq, dr, dt = torch.empty_like(q), torch.empty_like(rot), torch.empty_like(trans)
        s = torch.cuda.current_stream(dev).cuda_stream
        nv.call("hrp_fk_project_rot_bwd", ctx.robot.chain_on(dev).data_ptr(), q.data_ptr(), rot.data_ptr(), rot.shape[1], trans.data_ptr(),
                Kc.data_ptr() if ctx.hasK else None, B, ctx.root, gx.data_ptr() if gx is not None else None,
                gu.data_ptr() if gu is not None else None, dq.data_ptr(), dr.data_ptr(), dt.data_ptr(), s)
        return None, dq, dr, dt, None, None, None


class _SilhouetteFn(torch.autograd.Function):
    """posed mesh -> soft silhouette (hrp_mesh_pose + hrp_silhouette_fwd); backward through the rasteriser, the projection and the
    camera pose to (b2c_rot, b2c_trans) - the joint angles are detached on this path as in the reference (urdf_robot.py:267)."""

    @staticmethod
    def forward(ctx, robot, q, rot, trans, verts, vert_link, faces, K, H, W, root, sigma, blur, check_cap=False):
        xyz, uv = robot.pose_mesh(q, rot, trans, verts, vert_link, root=root, K=K)
        dev = q.device
        B, V = xyz.shape[0], xyz.shape[1]
        fc = faces.contiguous().to(torch.int32)
        alpha = torch.empty(B, H, W, device=dev)
        logp = torch.empty(B, H, W, dtype=torch.int64, device=dev)
        d = nv.SilhouetteDesc()
        count = torch.empty(B, H, W, dtype=torch.int32, device=dev) if check_cap else None
        d.count = count.data_ptr() if check_cap else None
        d.uv, d.xyz, d.faces = uv.data_ptr(), xyz.data_ptr(), fc.data_ptr()
        d.B, d.V, d.F, d.H, d.W = B, V, fc.shape[0], H, W
        d.sigma, d.blur_radius = sigma, blur
        d.alpha, d.logp = alpha.data_ptr(), logp.data_ptr()
        nv.call("hrp_silhouette_fwd", C.byref(d), torch.cuda.current_stream(dev).cuda_stream)
        if check_cap:
            robot.last_faces_per_pixel = int(count.max())          # (a device sync: a verification mode, off by default)
            d.count = None
            if robot.last_faces_per_pixel > 100:
                raise NotImplementedError(f"render_silhouette: {robot.last_faces_per_pixel} faces are kept at one pixel; pytorch3d would "
                                          "keep the 100 nearest (faces_per_pixel = 100, mesh_renderer.py:99) - that selection is not built")
        ctx.keep = (robot, q.detach(), rot.detach(), trans.detach(), verts, vert_link, fc, K.contiguous().float(), xyz, uv, logp, alpha, d, root)
        return alpha

    @staticmethod
    def backward(ctx, g_alpha):
        robot, q, rot, trans, verts, vert_link, fc, K, xyz, uv, logp, alpha, d, root = ctx.keep
        dev = q.device
        s = torch.cuda.current_stream(dev).cuda_stream
        B, V = xyz.shape[0], xyz.shape[1]
        ga = g_alpha.contiguous().float()
        d_uv = torch.empty(B, V, 2, device=dev)
        nv.call("hrp_silhouette_bwd", C.byref(d), ga.data_ptr(), d_uv.data_ptr(), s)
        d_xyz = torch.empty(B, V, 3, device=dev)
        nv.call("hrp_project_bwd", K.data_ptr(), xyz.data_ptr(), d_uv.data_ptr(), B, V, d_xyz.data_ptr(), s)
        chain, names = robot.mesh_chain_on(dev)
        root_kp = names.index(robot.link_names[root]) if root != 0 else -1
        qq, rr, tt = [x.contiguous().float() for x in (q, rot, trans)]
        d_rot, d_trans = torch.empty_like(rr), torch.empty_like(tt)
        nv.call("hrp_mesh_pose_bwd", chain.data_ptr(), qq.data_ptr(), rr.data_ptr(), tt.data_ptr(), B, root_kp,
                verts.contiguous().float().data_ptr(), vert_link.contiguous().to(torch.uint8).data_ptr(), V, d_xyz.data_ptr(),
                d_rot.data_ptr(), d_trans.data_ptr(), s)
        return None, None, d_rot, d_trans, None, None, None, None, None, None, None, None, None, None


class URDFRobot:
    def __init__(self, robot_type, urdf_path=None):
        if robot_type not in LINK_NAMES:
            raise NotImplementedError(f"robot '{robot_type}' has no keypoint table in this build")
        self.robot_type = robot_type
        self.urdf_path = urdf_path or _default_urdf(robot_type)
        self.actuated_joint_names = JOINT_NAMES[robot_type]
        self.global_scale = 1.0
        self.link_names, offsets = self.get_link_names_and_offsets()
        self.offsets = torch.as_tensor(offsets, dtype=torch.float32).reshape(1, len(self.link_names), 3, 1)
        self.chain, names = parse_chain(self.urdf_path, self.link_names, offsets)
        if names != list(self.actuated_joint_names):
            raise ValueError(f"URDF actuated joints {names} differ from JOINT_NAMES {self.actuated_joint_names}")
        self.dof = self.chain.dof
        self.nkp = len(self.link_names)
        self._dev = {}

    def get_link_names_and_offsets(self):
        """(keypoint link names, [nkp,3] offsets in those links' frames), reference urdf_robot.py:52-74."""
        if self.robot_type in ("panda", "kuka"):  # keypoints at the link origins
            names = LINK_NAMES[self.robot_type]
            return names, np.zeros((len(names), 3))
        if self.robot_type == "baxter":  # keypoints at joint origins, carried by the joints' parent links
            joints = {n.attrib["name"]: n for n in ET.parse(self.urdf_path).getroot().findall("joint")}
            names, offsets = [], []
            for jn in BAXTER_KEYPOINT_JOINTS:
                o = joints[jn].find("origin")
                offsets.append([float(v) for v in o.attrib.get("xyz", "0 0 0").split()] if o is not None else [0.0, 0.0, 0.0])
                names.append(joints[jn].find("parent").attrib["link"])
            return names, np.array(offsets)
        raise NotImplementedError(self.robot_type)

    def chain_on(self, device):
        key = str(device)
        if key not in self._dev:
            if device.type != "cuda":
                raise nv.HrpError("URDFRobot kinematics run on the GPU only (no CPU path)")
            self._dev[key] = torch.frombuffer(bytearray(bytes(self.chain)), dtype=torch.uint8).to(device)
        return self._dev[key]

    def _identity_cam(self, q):
        B = q.shape[0]
        rot = torch.tensor([1.0, 0, 0, 0, 1, 0], device=q.device).repeat(B, 1)
        return rot, torch.zeros(B, 3, device=q.device)

    def get_keypoints(self, jointcfgs, b2c_rot, b2c_trans):
        if b2c_rot.shape[1] not in (6, 4):
            raise NotImplementedError("rotation representations: 6-D (two matrix rows) and quaternion; rot9d is not built")
        return _FKFn.apply(self, jointcfgs, b2c_rot, b2c_trans, None, 0, ("xyz",))[0]

    def get_keypoints_root(self, jointcfgs, b2c_rot, b2c_trans, root=0):
        if root == 0:
            return self.get_keypoints(jointcfgs, b2c_rot, b2c_trans)
        assert 0 < root < len(self.link_names)
        if b2c_rot.shape[1] not in (6, 4):
            raise NotImplementedError("rotation representations: 6-D (two matrix rows) and quaternion; rot9d is not built")
        return _FKFn.apply(self, jointcfgs, b2c_rot, b2c_trans, None, root, ("xyz",))[0]

    def get_keypoints_and_projection(self, jointcfgs, b2c_rot, b2c_trans, K, root=0):
        """Fused extra: camera-frame keypoints AND their pixel projection in one launch."""
        return _FKFn.apply(self, jointcfgs, b2c_rot, b2c_trans, K, root, ("xyz", "uv"))

    def get_rotation_at_specific_root(self, jointcfgs, b2c_rot, b2c_trans, root=0):
        if root == 0:
            return b2c_rot
        assert root < len(self.link_names), (root, len(self.link_names))
        with torch.no_grad():
            return _FKFn.apply(self, jointcfgs, b2c_rot, b2c_trans, None, root, ("rot",))[0]

    def mesh_chain_on(self, device):
        """(device hrp_fk_chain whose key-point table lists the visual-mesh links, their names): reference urdf_robot.py:209-219."""
        key = "mesh:" + str(device)
        if key not in self._dev:
            if device.type != "cuda":
                raise nv.HrpError("URDFRobot kinematics run on the GPU only (no CPU path)")
            names = MESH_LINKS[self.robot_type]
            ch, _ = parse_chain(self.urdf_path, names)
            self._dev[key] = (torch.frombuffer(bytearray(bytes(ch)), dtype=torch.uint8).to(device), names)
        return self._dev[key]

    def pose_mesh(self, jointcfgs, b2c_rot, b2c_trans, verts, vert_link, root=0, K=None):
        """Camera-frame vertices of the posed robot mesh, one launch for the batch (reference: RobotMeshRenderer.get_robot_mesh
        per sample on the CPU, lib/utils/mesh_renderer.py:126-173, then the camera pose of
        get_rendered_mask_single_image_at_specific_root, urdf_robot.py:242-275 - re-rooted at key-point `root`, mirrored when
        the translation has negative depth).  verts [V, 3]: vertices in their links' frames; vert_link [V] uint8: index of the
        vertex's link in MESH_LINKS[robot_type].  -> xyz [B, V, 3] (and uv [B, V, 2] with K [B, 3, 3]).  No gradient: the
        trainer detaches the joint angles here."""
        dev = jointcfgs.device
        chain, names = self.mesh_chain_on(dev)
        root_kp = -1
        if root != 0:
            if self.link_names[root] not in names:
                raise NotImplementedError(f"key-point link {self.link_names[root]} is not a mesh link")
            root_kp = names.index(self.link_names[root])
        B, V = jointcfgs.shape[0], verts.shape[0]
        q, r, t = [x.detach().contiguous().float() for x in (jointcfgs, b2c_rot, b2c_trans)]
        vv, vl = verts.contiguous().float(), vert_link.contiguous().to(torch.uint8)
        xyz = torch.empty(B, V, 3, device=dev)
        uv = torch.empty(B, V, 2, device=dev) if K is not None else None
        Kc = K.contiguous().float() if K is not None else None
        nv.call("hrp_mesh_pose", chain.data_ptr(), q.data_ptr(), r.data_ptr(), t.data_ptr(), B, root_kp, vv.data_ptr(), vl.data_ptr(), V,
                Kc.data_ptr() if Kc is not None else None, xyz.data_ptr(), uv.data_ptr() if uv is not None else None,
                torch.cuda.current_stream(dev).cuda_stream)
        return xyz if K is None else (xyz, uv)

    def _check_mesh(self, verts, vert_link, faces):
        """Index ranges of a mesh tuple, once per tuple (the kernels index vertices by face entry and link poses by
        vert_link without bounds checks; a malformed .obj would read out of bounds).  One device sync per new mesh."""
        key = (faces.data_ptr(), vert_link.data_ptr(), int(verts.shape[0]), int(faces.shape[0]))
        seen = self.__dict__.setdefault("_mesh_checked", set())
        if key in seen:
            return
        V = int(verts.shape[0])
        nlinks = len(MESH_LINKS[self.robot_type])
        if faces.numel() and (int(faces.min()) < 0 or int(faces.max()) >= V):
            raise ValueError(f"mesh faces index vertices outside [0, {V})")
        if vert_link.numel() != V or (V and int(vert_link.max()) >= nlinks):
            raise ValueError(f"mesh vert_link must hold one link index < {nlinks} per vertex")
        seen.add(key)

    def render_silhouette(self, jointcfgs, b2c_rot, b2c_trans, mesh, K, image_size, root=0, sigma=1e-8, blur_radius=None,
                          check_faces_per_pixel=False):
        """Soft silhouettes [B, H, W] of the posed robot mesh for a whole batch - the loop of scripts/train_sim2real.py:415-418 over
        get_rendered_mask_single_image_at_specific_root (urdf_robot.py:242-275) with the renderer of
        lib/utils/mesh_renderer.py:78-109 (pytorch3d MeshRasterizer + SoftSilhouetteShader, sigma 1e-8, blur_radius
        log(1 / 1e-4 - 1) * sigma).  mesh = (verts [V, 3] in link frames, vert_link [V], faces [F, 3]); K [B, 3, 3] the intrinsics of
        the rendered image (set_robot_renderer scales K_original by 0.5), image_size = (H, W).  Gradients reach b2c_rot / b2c_trans.
        PARITY UNPINNED (csrc/silhouette.hip): pytorch3d is not available to the build; the algorithm is restated from its
        published form and checked against this repository's own torch restatement (oracle/silhouette.py).
        check_faces_per_pixel: verify (one device sync) that no pixel keeps more than pytorch3d's faces_per_pixel = 100 faces - the
        only place where this rasteriser could differ from the published algorithm; raises NotImplementedError otherwise and leaves
        the maximum in self.last_faces_per_pixel."""
        if b2c_rot.shape[1] != 6:
            raise NotImplementedError("render_silhouette: 6-D rotations only")
        verts, vert_link, faces = mesh
        self._check_mesh(verts, vert_link, faces)
        if blur_radius is None:
            blur_radius = float(np.log(1.0 / 1e-4 - 1.0) * sigma)
        H, W = image_size
        return _SilhouetteFn.apply(self, jointcfgs, b2c_rot, b2c_trans, verts, vert_link, faces, K, int(H), int(W), root,
                                   float(sigma), float(blur_radius), bool(check_faces_per_pixel))

    def set_robot_renderer(self, K_original, original_image_size=(480, 640), scale=0.5, device="cuda", mesh=None):
        """reference urdf_robot.py:201-227: intrinsics and image size scaled by `scale`, the visual meshes of MESH_LINKS next to the
        URDF (`meshes/visual/<link>/<link>.obj`) unless `mesh` = (verts, vert_link, faces) is given."""
        from hrpe_amd.lib.utils.mesh_renderer import RobotMeshRenderer
        fx, fy = float(K_original[0, 0]) * scale, float(K_original[1, 1]) * scale
        cx, cy = float(K_original[0, 2]) * scale, float(K_original[1, 2]) * scale
        size = (int(original_image_size[0] * scale), int(original_image_size[1] * scale))
        files = None
        if mesh is None:
            base = os.path.dirname(self.urdf_path)
            short = [n.replace("panda_", "") for n in MESH_LINKS[self.robot_type]]
            files = [os.path.join(base, "meshes", "visual", n, n + ".obj") for n in short]
        return RobotMeshRenderer([-fx, -fy], [cx, cy], size, robot=self, mesh_files=files, device=device, mesh=mesh)

    def get_rendered_masks(self, joint_angles, rot, trans, renderer, root=0):
        """The loop of scripts/train_sim2real.py:415-418 for the whole batch: [B, H, W] rendered masks, differentiable in rot / trans
        (the joint angles are detached, urdf_robot.py:267)."""
        return renderer.silhouettes(self, joint_angles.detach(), rot, trans, root=root)

    def get_rendered_mask_single_image_at_specific_root(self, joint_angles, rot, trans, robot_mesh, robot_renderer_gpu, root=0):
        """reference urdf_robot.py:259-275 (one sample; `robot_mesh` is not needed: the renderer poses the mesh itself) -> [1, H, W]"""
        return self.get_rendered_masks(joint_angles[None], rot[None], trans[None], robot_renderer_gpu, root=root)

    def get_keypoints_only_fk(self, jointcfgs):
        rot, tr = self._identity_cam(jointcfgs)
        return _FKFn.apply(self, jointcfgs, rot, tr, None, 0, ("xyz",))[0]

    def get_keypoints_only_fk_at_specific_root(self, jointcfgs, root=0):
        if root == 0:
            return self.get_keypoints_only_fk(jointcfgs)
        assert 0 < root < len(self.link_names)
        rot, tr = self._identity_cam(jointcfgs)
        return _FKFn.apply(self, jointcfgs, rot, tr, None, root, ("xyz",))[0]
