"""Soft-argmax ("integral") pose head with the reference's interface (lib/utils/integral.py:75-189).

``HeatmapIntegralPose(backbone, **kwargs).forward(out, root_trans=, K=)`` -> ``(uvd, xyz)``.  The HRNet
branch (integral.py:147-186) runs as ONE pass of the online soft-argmax kernel over the logits plus the
per-sample geometry kernel; inside ``RootNetwithRegInt`` the same kernels are used without the NCHW
round trip."""
import torch

from hrpe_amd.runtime import PlannedModule


class HeatmapIntegralPose(PlannedModule):
    def __init__(self, backbone, **kwargs):
        super().__init__()
        self.backbone_name = backbone
        self.norm_type = kwargs["norm_type"]
        if self.norm_type != "softmax":
            raise NotImplementedError(self.norm_type)
        self.num_joints = kwargs["num_joints"]
        self.depth_dim = kwargs["depth_dim"]
        self.height_dim = kwargs["height_dim"]
        self.width_dim = kwargs["width_dim"]
        self.rootid = kwargs.get("rootid", 0)
        self.fixroot = kwargs.get("fixroot", False)
        bbox = kwargs.get("bbox_3d_shape", (2300, 2300, 2300))
        self.bbox_3d_shape = torch.tensor(bbox).float()
        self.depth_factor = float(self.bbox_3d_shape[2]) * 1e-3
        self.image_size = kwargs["image_size"]
        # integral.py:105-145 (ResNet) and :147-186 (HRNet) compute the same soft-argmax - the ResNet branch divides
        # the softmax by its own sum (= 1) once more; one kernel serves both (checked against the reference for both)
        if backbone not in ("hrnet", "hrnet32", "hrnet48", "resnet", "resnet34", "resnet50"):
            raise NotImplementedError(f"soft-argmax head for backbone {backbone!r}")

    def emit(self, pb, heat, z_root_dense, Kmat):
        """heat NHWC logits; z_root [N,1] dense metres; K [N,9] -> (uvd, xyz) dense fp32."""
        J = self.num_joints
        uvd = pb.softargmax(heat, J, self.depth_dim, self.rootid, self.fixroot)
        ones = pb.constant(heat.N, 1, 1000.0)
        _, xyz, _, _ = pb.pose_geometry(z_root_dense, ones, uvd, Kmat, J, self.rootid, self.image_size,
                                        self.depth_factor)
        return uvd, xyz

    def forward(self, out, flip_test=False, **kwargs):
        K, root_trans = kwargs["K"], kwargs["root_trans"]
        z = root_trans[:, 2:3].to(out.device)
        uvd, xyz = self._run(out, z, K.to(out.device).reshape(-1, 9))
        B = out.shape[0]
        return uvd.view(B, self.num_joints, 3), xyz.view(B, self.num_joints, 3)

    def _build(self, pb, out, z, K):
        N, Cc, H, W = out.shape
        t = pb.image_input("out", N, Cc, H, W)
        t.requires_grad = pb.plan.need_grad and out.requires_grad
        zt = pb.vector_input("z", N, 1, dense=True)
        zt.requires_grad = pb.plan.need_grad and z.requires_grad
        Kt = pb.vector_input("K", N, 9, dense=True)
        uvd, xyz = self.emit(pb, t, zt, Kt)
        J = self.num_joints
        return ["out", "z", "K"], [("dense", uvd, (N, J * 3)), ("dense", xyz, (N, J * 3))], {"out": t}
