"""Full pose network on the plan runtime - drop-in for reference lib/models/full_net.py:18-435.

``RootNetwithRegInt(init_param_dict, args).forward(x_reg_input, x_root_input, k_value, K, ...)`` returns the
reference's 8-tuple ``(pred_pose, pred_rot, pred_trans, pred_root_uv, pred_depth, pred_uvd, pred_xyz_int,
pred_xyz_fk)``.  One static plan holds both HRNet-W32 trunks, the depth head, the one-pass soft-argmax,
the camera geometry, the iterative joint / rotation regressors and the FK kernel; nothing leaves the GPU
(the reference copies the depth to the host and back, full_net.py:249, 287)."""
import math
import os
import time

import numpy as np
import torch
import torch.nn as nn

from hrpe_amd.lib.dataset.const import JOINT_NAMES
from hrpe_amd.lib.utils.geometries import rotmat_to_quat, rotmat_to_rot6d
from hrpe_amd.lib.utils.integral import HeatmapIntegralPose
from hrpe_amd.lib.utils.urdf_robot import URDFRobot
from hrpe_amd.runtime import PlannedModule
from hrpe_amd.plan import Term
from .backbones import HRnet
from .backbones.HRnet import BatchNorm1d, BatchNorm2d, Conv2d, emit_trunks, get_hrnet
from .backbones.Resnet import _StemConv, get_resnet

# the two iterative regressors (joint angles, rotation) as ONE chain of fused launches (PlanBuilder.regressors, csrc/regressor.hip);
# False (tests, A/B measurements): the round-5 form, one launch per nn.Linear / cat / dropout
FUSED_REGRESSORS = True
# soft-argmax + camera geometry and the regressors as two lanes of a parallel block of their own (_build)
HEADS_IN_LANES = os.environ.get("HRP_HEADS_IN_LANES", "1") not in ("0", "")
_RESNETS = ["resnet", "resnet34", "resnet50", "resnet101"]
_HRNETS = ["hrnet", "hrnet32"]


class Linear(PlannedModule):
    """[out, in] weight + bias container (torch.nn.Linear's initialisation); executed by the skinny fp32 GEMM kernels
    hrp_linear_* straight from the PyTorch-shaped weight."""

    def __init__(self, in_features, out_features):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.empty(out_features))
        bound = 1.0 / math.sqrt(in_features)
        nn.init.uniform_(self.weight, -bound, bound)
        nn.init.uniform_(self.bias, -bound, bound)

    def emit(self, pb, x, residual=None):
        return pb.linear(x, self.weight, self.bias, residual=residual)


class ConvTranspose2d(PlannedModule):
    """Parameter holder of nn.ConvTranspose2d(cin, cout, 4, stride 2, padding 1, bias=False) (full_net.py:196-203):
    weight [cin, cout, 4, 4]; executed by PlanBuilder.deconv4x4s2."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.weight = nn.Parameter(torch.empty(in_channels, out_channels, 4, 4))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))


class RootNetwithRegInt(PlannedModule):
    def __init__(self, init_param_dict, args, **kwargs):
        super().__init__()
        robot_type = init_param_dict["robot_type"]
        dims = {"panda": (8, 7), "kuka": (7, 8), "baxter": (15, 17)}
        if robot_type not in dims:
            raise ValueError(f"Robot type {robot_type} is not supported.")
        DoF, nkpt = dims[robot_type]
        npose = DoF
        self.robot = URDFRobot(robot_type)
        self.backbone_name = args.backbone_name
        self.rootnet_backbone_name = args.rootnet_backbone_name
        self.use_rpmg = args.use_rpmg
        self.n_iter = args.n_iter
        self.norm_type = "softmax"
        self.num_joints = nkpt
        self.image_size = args.other_image_size
        self.depth_dim = 64
        self.height_dim = int(self.image_size / 4)
        self.width_dim = int(self.image_size / 4)
        self.bbox_3d_shape = args.bbox_3d_shape
        self.reference_keypoint_id = args.reference_keypoint_id
        self.rotation_dim = args.rotation_dim
        self.p_dropout = args.p_dropout
        if self.backbone_name in _RESNETS:
            # the shipped full.yaml: ResNet trunk -> (global pool -> FC heads) and (3 x ConvTranspose2d+BN+ReLU -> 1x1)
            self.reg_backbone = get_resnet(self.backbone_name)
            self.feature_channel = self.reg_backbone.block.expansion * 512
            self.deconv_dim = [256, 256, 256]
            self.deconv_layers = self._make_deconv_layer()
            self.final_layer = Conv2d(self.deconv_dim[2], self.num_joints * self.depth_dim, 1, bias=True)
        elif self.backbone_name in _HRNETS:
            self.reg_backbone = get_hrnet(type_name=32, num_joints=self.num_joints, depth_dim=self.depth_dim,
                                          pretrain=True, generate_feat=True, generate_hm=True)
            self.feature_channel = 2048
        else:
            raise NotImplementedError
        self.integral_layer = HeatmapIntegralPose(
            backbone=self.backbone_name, num_joints=self.num_joints, depth_dim=self.depth_dim,
            height_dim=self.height_dim, width_dim=self.width_dim, norm_type=self.norm_type,
            image_size=self.image_size, bbox_3d_shape=self.bbox_3d_shape, rootid=self.reference_keypoint_id,
            fixroot=args.fix_root)
        self.reg_joint_map = args.reg_joint_map
        self.direct_reg_rot = args.direct_reg_rot
        self.rot_iterative_matmul = args.rot_iterative_matmul
        if self.reg_joint_map and self.backbone_name not in ("resnet34", "resnet50"):
            raise NotImplementedError("reg_joint_map reads the ResNet feature map (full_net.py:313-316) and its integral layer "
                                      "accepts resnet34 / resnet50 only (integral.py:206, 234)")
        if self.rotation_dim not in (6, 4):
            raise NotImplementedError("rotation_dim: 6 (two rows of the rotation matrix) or 4 (quaternion)")
        if self.rotation_dim != 6 and (self.direct_reg_rot or self.rot_iterative_matmul):
            raise NotImplementedError("direct_reg_rot / rot_iterative_matmul decode a 6-D rotation (full_net.py:127, 349)")
        if self.reg_joint_map:       # full_net.py:87-93, 218-237: 3 x (conv3x3 + BN + ReLU), a 1x1 conv to one map per joint
            from hrpe_amd.lib.dataset.const import JOINT_BOUNDS
            dims, mods, cin = list(args.joint_conv_dim), [], self.feature_channel
            for c in dims:
                mods += [Conv2d(cin, c, 3, bias=True), BatchNorm2d(c), nn.Identity()]
                cin = c
            self.joint_conv_dim, self.joint_conv_layers = dims, nn.Sequential(*mods)
            self.joint_final_layer = Conv2d(dims[2], npose, 1, bias=True)
            self.register_buffer("joint_bounds", torch.tensor(JOINT_BOUNDS[robot_type]).float(), persistent=False)
        else:
            self.fc_pose_1 = Linear(self.feature_channel + npose, 1024)
            self.fc_pose_2 = Linear(1024, 1024)
            self.decpose = Linear(1024, npose)
            nn.init.xavier_uniform_(self.decpose.weight, gain=0.01)
        if self.direct_reg_rot:      # full_net.py:108-115: six stacked layers on the feature, one skip, decrot -> 6
            self.fc_rot_1 = Linear(self.feature_channel, 1024)
            for i in range(2, 7):
                setattr(self, f"fc_rot_{i}", Linear(1024, 1024))
            self.decrot = Linear(1024, 6)
        else:
            self.fc_rot_1 = Linear(self.feature_channel + self.rotation_dim, 1024)
            self.fc_rot_2 = Linear(1024, 1024)
            self.decrot = Linear(1024, self.rotation_dim)
        nn.init.xavier_uniform_(self.decrot.weight, gain=0.01)
        if self.rootnet_backbone_name in _HRNETS:
            self.rootnet_backbone = get_hrnet(type_name=32, num_joints=nkpt, depth_dim=self.depth_dim,
                                              pretrain=True, generate_feat=True, generate_hm=False)
            self.inplanes = 2048
        elif self.rootnet_backbone_name in ["resnet", "resnet50", "resnet34"]:
            self.rootnet_backbone = get_resnet(self.rootnet_backbone_name)       # full_net.py:136-138
            self.inplanes = self.rootnet_backbone.block.expansion * 512
        else:
            raise NotImplementedError
        self.multi_kp = args.multi_kp
        self.add_fc = args.add_fc
        if self.add_fc:      # full_net.py:150-157 (depth_dropout is constructed there but never applied)
            from .depth_net import _Linear1x1
            self.depth_fc_d1 = _Linear1x1(self.inplanes, 1024)
            self.depth_fc_d2 = _Linear1x1(1024, 512)
            self.depth_bn = BatchNorm1d(512)                 # BatchNorm1d: the same parameters / buffers
            self.depth_fc_u2 = _Linear1x1(512, 1024)
            self.depth_fc_u1 = _Linear1x1(1024, self.inplanes)
        # full_net.py:146-148: with multi_kp the depth layer predicts one gamma per listed key-point
        self.kps_need_depth = list(args.kps_need_depth) if self.multi_kp else [args.reference_keypoint_id]
        self.depth_num = len(self.kps_need_depth)
        self.depth_layer = Conv2d(self.inplanes, self.depth_num, 1, bias=True)
        # reference full_net.py:167-177: every conv ~ N(0, sqrt(2/n)), BN = (1, 0), depth layer N(0, 0.001)
        for m in self.modules():
            if isinstance(m, Conv2d):
                n = m.kernel_size * m.kernel_size * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2.0 / n))
            elif isinstance(m, _StemConv):
                m.weight.data.normal_(0, math.sqrt(2.0 / (7 * 7 * 64)))
        nn.init.normal_(self.depth_layer.weight, std=0.001)
        nn.init.constant_(self.depth_layer.bias, 0)
        pose_params = init_param_dict["pose_params"]
        which = "mean" if init_param_dict["init_pose_from_mean"] else "zero"
        init_pose = torch.tensor([[pose_params[which][robot_type][k] for k in JOINT_NAMES[robot_type]]]).float()
        cam = np.array(init_param_dict["cam_params"])
        to_rot = rotmat_to_rot6d if self.rotation_dim == 6 else rotmat_to_quat        # full_net.py:186-189
        init_rot = to_rot(torch.from_numpy(cam[:3, :3]).unsqueeze(0).float()).float()
        self.register_buffer("init_pose", init_pose)
        self.register_buffer("init_rot", init_rot)

    def _make_deconv_layer(self):
        """full_net.py:194-216: keys deconv_layers.{0,3,6}.weight ([Cin, Cout, 4, 4]) and .{1,4,7}.* (BatchNorm)."""
        mods, cin = [], self.feature_channel
        for cout in self.deconv_dim:
            mods += [ConvTranspose2d(cin, cout), BatchNorm2d(cout), nn.Identity()]
            cin = cout
        return nn.Sequential(*mods)

    def _resnet_reg_units(self, pb, xs, out):
        """Generator over the regression path of the shipped full.yaml (full_net.py:293-298: x_out = trunk(x);
        xf = avgpool(x_out); heat-map = final_layer(deconv_layers(x_out))), one unit (stem, a residual block, a deconv
        layer) per step; emit_trunks advances it in a lane of its own next to the HRNet root trunk."""
        rb = self.reg_backbone
        y = pb.stem7x7_s2d(xs, rb.conv1.weight, want_stats=pb.plan.training)
        h = pb.maxpool3x3s2(pb.act([Term(y, rb.bn1)], relu=True))
        yield
        for layer in (rb.layer1, rb.layer2, rb.layer3, rb.layer4):
            for blk in layer:
                h = blk.emit(pb, h)
                yield
        out["xf"] = pb.avgpool(h)
        if self.reg_joint_map:       # full_net.py:313-316 + HeatmapIntegralJoint (integral.py:206-232)
            from .backbones.HRnet import conv_bn
            j = h
            for i in (0, 3, 6):
                j = pb.act([conv_bn(pb, j, self.joint_conv_layers[i], self.joint_conv_layers[i + 1])], relu=True)
            coord = pb.softargmax_flat(self.joint_final_layer.emit(pb, j), self.joint_bounds.shape[0])
            N, nj = coord.N, self.joint_bounds.shape[0]
            rng, lo = pb.constant(N, nj, 0.0), pb.constant(N, nj, 0.0)
            rng.buf.view(N, nj).copy_((self.joint_bounds[:, 1] - self.joint_bounds[:, 0]).to(rng.buf.device).expand(N, nj))
            lo.buf.view(N, nj).copy_(self.joint_bounds[:, 0].to(lo.buf.device).expand(N, nj))
            out["pose"] = pb.act([Term(pb.row_scale(coord, rng)), Term(lo)], relu=False)     # joints = coord * range + lower bound
            yield
        for i in range(0, len(self.deconv_layers), 3):
            y = pb.deconv4x4s2(h, self.deconv_layers[i].weight, want_stats=pb.plan.training)
            h = pb.act([Term(y, self.deconv_layers[i + 1])], relu=True)
            yield
        out["heat"] = self.final_layer.emit(pb, h)

    def _resnet_root_units(self, pb, xs, out):
        """Generator over a ResNet root trunk (full_net.py:262-266: global average pool of the feature map), one unit
        per step, for emit_trunks' rider lane."""
        rb = self.rootnet_backbone
        y = pb.stem7x7_s2d(xs, rb.conv1.weight, want_stats=pb.plan.training)
        h = pb.maxpool3x3s2(pb.act([Term(y, rb.bn1)], relu=True))
        yield
        for layer in (rb.layer1, rb.layer2, rb.layer3, rb.layer4):
            for blk in layer:
                h = blk.emit(pb, h)
                yield
        out["feat"] = pb.avgpool(h)

    def _iter_head(self, pb, xf, init, np_, fc1, fc2, dec, matmul=False):
        """full_net.py:318-331: p <- p + dec(drop(fc2(drop(fc1([xf; p])))))  x n_iter; matmul (rot_iterative_matmul,
        :346-362): p <- rot6d(R(dec(..)) @ R(p)) instead of the sum."""
        pred = init          # [N, np_] plan input: the module's init buffer expanded, or the caller's per-sample start (full_net.py:245-248)
        for _ in range(self.n_iter):
            xc = pb.cat_cols([xf, pred])
            h = pb.dropout(fc1.emit(pb, xc), self.p_dropout)
            h = pb.dropout(fc2.emit(pb, h), self.p_dropout)
            if matmul:
                pred = pb.rot6d_compose(pb.dense(dec.emit(pb, h)), pb.dense(pred))
            else:
                pred = dec.emit(pb, h, residual=pred)
        return pred

    def _unfused_heads(self, pb, xf, res, ip, ir):
        """The regressors one launch per layer (round 5; the variants the fused chain does not cover: reg_joint_map,
        direct_reg_rot, rot_iterative_matmul).  The pose and the rotation regressor are independent chains of 12 small GEMMs
        each: two lanes; each gets a private copy of the feature so that its gradient accumulates lane-locally."""
        if self.reg_joint_map:
            pose, xf_rot = res["pose"], xf
        else:
            xf_pose, xf_rot = pb.new_like(xf), pb.new_like(xf)
            pb.copy_cols(xf, xf_pose)
            pb.copy_cols(xf, xf_rot)
        with (pb.parallel(2) if not self.reg_joint_map else _NoBlock()) as par:
            if not self.reg_joint_map:
                with par.lane(0):
                    pose = self._iter_head(pb, xf_pose, ip, self.init_pose.shape[1], self.fc_pose_1,
                                           self.fc_pose_2, self.decpose)
            with par.lane(1):
                if self.direct_reg_rot:      # full_net.py:333-345
                    xc1 = self.fc_rot_1.emit(pb, xf_rot)
                    xc = xc1
                    for i in range(2, 6):
                        xc = getattr(self, f"fc_rot_{i}").emit(pb, xc)
                    rot = self.decrot.emit(pb, self.fc_rot_6.emit(pb, xc, residual=xc1))
                else:
                    rot = self._iter_head(pb, xf_rot, ir, self.rotation_dim, self.fc_rot_1, self.fc_rot_2,
                                          self.decrot, matmul=self.rot_iterative_matmul)
        return pose, rot

    def _depth_gamma(self, pb, feat):
        """depth_layer on the pooled root feature (full_net.py:271-274); add_fc: the hour-glass MLP of :261-270 first
        (fc_d1 -> fc_d2 -> BatchNorm1d -> LeakyReLU -> fc_u2, 0.5 (. + d1), fc_u1, 0.5 (. + feature))."""
        if self.add_fc:
            N = feat.N
            half = lambda t: pb.row_scale(pb.dense(t), pb.constant(N, t.C, 0.5))      # noqa: E731
            f1 = pb.conv(feat, self.depth_fc_d1.weight, self.depth_fc_d1.bias)
            f2 = pb.conv(f1, self.depth_fc_d2.weight, self.depth_fc_d2.bias, want_stats=pb.plan.training)
            mid = pb.act([Term(f2, self.depth_bn)], relu="leaky")
            f3 = half(pb.conv(mid, self.depth_fc_u2.weight, self.depth_fc_u2.bias, residual=f1))
            feat = half(pb.conv(f3, self.depth_fc_u1.weight, self.depth_fc_u1.bias, residual=feat))
        return pb.dense(self.depth_layer.emit(pb, feat))

    def _build(self, pb, x_reg, x_root, k_value, K, init_pose, init_rot):
        N = x_reg.shape[0]
        J, root = self.num_joints, self.reference_keypoint_id
        resnet_reg = self.backbone_name in _RESNETS
        resnet_root = self.rootnet_backbone_name not in _HRNETS
        lazy = lambda resnet: {} if resnet else {"lazy": True}      # noqa: E731  (HRNet stems read the image with pb.conv)
        xr = (pb.image_input_s2d if resnet_reg else pb.image_input)("x_reg", N, 3, x_reg.shape[2], x_reg.shape[3], u8=x_reg.dtype == torch.uint8,
                                                                    **lazy(resnet_reg))
        if not resnet_root and HRnet.TRUNK_FP32_FROM == "1":
            # the WHOLE DepthNet in fp32 inside a bf16 plan (measurement mode, DESIGN 4: the cheapest mode that keeps every
            # key-point of the fixture within 0.5 px - bf16 operands anywhere in this trunk do not)
            xo = pb.image_input("x_root", N, 3, x_root.shape[2], x_root.shape[3], u8=x_root.dtype == torch.uint8, dtype=torch.float32, lazy=True)
        else:
            xo = (pb.image_input_s2d if resnet_root else pb.image_input)("x_root", N, 3, x_root.shape[2], x_root.shape[3], u8=x_root.dtype == torch.uint8,
                                                                         **lazy(resnet_root))
        kv = pb.vector_input("k_value", N, 1, dense=True)
        Km = pb.vector_input("K", N, 9, dense=True)
        ip = None if self.reg_joint_map else pb.vector_input("init_pose", N, self.init_pose.shape[1], dense=True)
        ir = None if self.direct_reg_rot else pb.vector_input("init_rot", N, self.rotation_dim, dense=True)
        # The trunks share nothing until pose_geometry.  HRNet trunks are emitted in lockstep (emit_trunks: one flat
        # parallel block per step, a lane per branch); a ResNet chain rides along those blocks as one more lane.
        res, rootd = {}, {}
        reg_units = self._resnet_reg_units(pb, xr, res) if resnet_reg else None
        root_units = self._resnet_root_units(pb, xo, rootd) if resnet_root else None
        if resnet_reg and resnet_root:
            with pb.parallel(2) as par:
                with par.lane(0):
                    for _ in reg_units:
                        pass
                with par.lane(1):
                    for _ in root_units:
                        pass
            heat, xf, feat_root = res["heat"], res["xf"], rootd["feat"]
            gamma = self._depth_gamma(pb, feat_root)
        elif resnet_reg:
            (ys_root,) = emit_trunks(pb, [self.rootnet_backbone], [xo], rider=reg_units)
            with pb.parallel(2) as par:
                with par.lane(0):
                    for _ in reg_units:      # whatever the trunk blocks did not reach
                        pass
                    heat, xf = res["heat"], res["xf"]
                with par.lane(1):
                    _, feat_root = self.rootnet_backbone.emit_heads(pb, ys_root)
                    gamma = self._depth_gamma(pb, feat_root)
        elif resnet_root:
            (ys_reg,) = emit_trunks(pb, [self.reg_backbone], [xr], rider=root_units)
            with pb.parallel(2) as par:
                with par.lane(0):
                    heat, xf = self.reg_backbone.emit_heads(pb, ys_reg)
                with par.lane(1):
                    for _ in root_units:
                        pass
                    gamma = self._depth_gamma(pb, rootd["feat"])
        else:
            ys_reg, ys_root = emit_trunks(pb, [self.reg_backbone, self.rootnet_backbone], [xr, xo])
            with pb.parallel(2) as par:
                with par.lane(0):
                    heat, xf = self.reg_backbone.emit_heads(pb, ys_reg)
                with par.lane(1):
                    _, feat_root = self.rootnet_backbone.emit_heads(pb, ys_root)
                    gamma = self._depth_gamma(pb, feat_root)
        depths = None
        if self.multi_kp:      # full_net.py:275-279: pred_depths = gamma * k / 1000 per listed key-point; the root's feeds the rest
            from hrpe_amd.plan import TensorH
            gamma_all = gamma
            kvn = pb.cat_cols([kv] * self.depth_num)
            depths = pb.row_scale(pb.row_scale(gamma_all, kvn), pb.constant(N, self.depth_num, 1e-3))
            ri = self.kps_need_depth.index(root)
            col = TensorH(pb.plan, N, 1, 1, 1, torch.float32, buf=gamma_all.buf, offset=gamma_all.offset + ri,
                          pitch=gamma_all.pitch, base=gamma_all.base if gamma_all.base is not None else gamma_all)
            col.requires_grad = gamma_all.requires_grad
            gamma = pb.plan.new(N, 1, 1, 1, torch.float32, pitch=1)
            pb.copy_cols(col, gamma)
        il = self.integral_layer
        fused = (FUSED_REGRESSORS and not self.reg_joint_map and not self.direct_reg_rot and not self.rot_iterative_matmul
                 and self.fc_pose_2.weight.shape[0] % 16 == 0 and xf.C % 4 == 0)
        # Between the trunks and the forward kinematics two chains share nothing: heat-map -> soft-argmax -> camera geometry, and
        # feature -> regressors.  Both sit where the trunks have joined and nothing else can overlap them, so they get a lane each
        # (HEADS_IN_LANES; forward 0.12 + 0.15 ms one after the other, backward 0.11 + 0.24).
        with (pb.parallel(2) if HEADS_IN_LANES else _NoBlock()) as par:
            with par.lane(0):
                uvd = pb.softargmax(heat, J, il.depth_dim, root, il.fixroot)
                depth, xyz_int, root_uv, trans = pb.pose_geometry(gamma, kv, uvd, Km, J, root, self.image_size,
                                                                  il.depth_factor)
            with par.lane(1):
                if fused:
                    # both loops of full_net.py:318-331 / :365-378 in one chain: 1 + 1 + n_iter + 1 launches forward, n_iter + 2
                    # backward, the xf part of fc_*_1 hoisted out of the loop
                    pose, rot = pb.regressors(xf, [(ip, self.fc_pose_1, self.fc_pose_2, self.decpose),
                                                   (ir, self.fc_rot_1, self.fc_rot_2, self.decrot)], self.n_iter, self.p_dropout)
                else:
                    pose, rot = self._unfused_heads(pb, xf, res, ip, ir)
                pose_d, rot_d = pb.dense(pose), pb.dense(rot)
        xyz_fk, _, _ = pb.fk(self.robot.chain_on(pb.plan.device), self.robot.dof, self.robot.nkp, pose_d, rot_d, trans, root)
        outs = [("dense", pose_d, (N, pose_d.C)), ("dense", rot_d, (N, rot_d.C)), ("dense", trans, (N, 3)),
                ("dense", root_uv, (N, 2)), ("dense", depth, (N, 1)), ("dense", uvd, (N, J, 3)),
                ("dense", xyz_int, (N, J, 3)), ("dense", xyz_fk, (N, J, 3))]
        if depths is not None:      # full_net.py:392-395: the 9-tuple carries pred_depths after pred_depth
            outs.insert(5, ("dense", depths, (N, self.depth_num)))
        return ["x_reg", "x_root", "k_value", "K", "init_pose", "init_rot"], outs, {"x_reg": xr, "x_root": xo}

    def forward(self, x_reg_input, x_root_input, k_value, K, init_pose=None, init_rot=None, test_fps=False):
        dev = x_reg_input.device
        B = x_reg_input.shape[0]
        # full_net.py:245-248: the iterative regressors start from the module's buffers unless the caller brings a start per sample
        init_pose = (self.init_pose.expand(B, -1) if init_pose is None else init_pose).to(dev).float().reshape(B, -1).contiguous()
        init_rot = (self.init_rot.expand(B, -1) if init_rot is None else init_rot).to(dev).float().reshape(B, -1).contiguous()
        if torch.is_grad_enabled() and (init_pose.requires_grad or init_rot.requires_grad):
            # the reference backpropagates through the regressors to a caller-supplied start (full_net.py:245-248, 318-331); the plan treats
            # the starts as inputs without gradient - say so instead of silently returning None for them
            raise NotImplementedError("RootNetwithRegInt: init_pose / init_rot that require grad are not differentiated; pass them detached")
        if init_pose.shape[1] != self.init_pose.shape[1] or init_rot.shape[1] != self.rotation_dim:
            raise ValueError(f"init_pose / init_rot must be [B, {self.init_pose.shape[1]}] / [B, {self.rotation_dim}]")
        if test_fps:
            torch.cuda.synchronize(dev)
            t0 = time.time()
        outs = self._run(x_reg_input, x_root_input, k_value.to(dev).reshape(-1, 1), K.to(dev).reshape(-1, 9), init_pose, init_rot)
        if test_fps:
            torch.cuda.synchronize(dev)
            t = time.time() - t0
            # full_net.py:253-286, 385-392 time the root-depth part and the rest one after the other.  Here both trunks
            # run concurrently inside one plan, so the root part is timed as what it is on its own - the same modules
            # emitted as a plan of their own (root trunk + depth layer) - and the rest is the remainder of the whole.
            if getattr(self, "_fps_root", None) is None:
                object.__setattr__(self, "_fps_root", _RootOnly(self))
            self._fps_root.train(False)       # (an inference plan: no second running-statistics update in train mode)
            object.__setattr__(self._fps_root, "_compute_dtype", self._compute_dtype)
            with torch.no_grad():
                self._fps_root(x_root_input, k_value.to(dev).reshape(-1, 1))      # builds the plan on first use
                torch.cuda.synchronize(dev)
                t1 = time.time()
                self._fps_root(x_root_input, k_value.to(dev).reshape(-1, 1))
                torch.cuda.synchronize(dev)
            t_root = min(time.time() - t1, t)
            return outs + ((t_root, t - t_root, t),)
        return outs


class _NoBlock:
    """Stand-in for a parallel block when only one of its chains exists (reg_joint_map: no iterative pose head): the
    remaining chain is emitted into the current lane."""

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def lane(self, i):
        return self


class _RootOnly(PlannedModule):
    """The root-depth part of RootNetwithRegInt (full_net.py:252-287) as a plan of its own, for the test_fps timers: shares
    the parent's modules without registering them a second time."""

    def __init__(self, parent):
        super().__init__()
        object.__setattr__(self, "_parent", parent)

    def parameters(self, recurse=True):
        return iter(list(self._parent.rootnet_backbone.parameters()) + list(self._parent.depth_layer.parameters()))

    def _build(self, pb, x_root, k_value):
        P = self._parent
        N = x_root.shape[0]
        kv = pb.vector_input("k_value", N, 1, dense=True)
        if P.rootnet_backbone_name in _HRNETS:
            t = pb.image_input("x_root", N, 3, x_root.shape[2], x_root.shape[3], u8=x_root.dtype == torch.uint8)
            _, feat = P.rootnet_backbone.emit(pb, t)
        else:
            t = pb.image_input_s2d("x_root", N, 3, x_root.shape[2], x_root.shape[3], u8=x_root.dtype == torch.uint8)
            feat = pb.avgpool(P.rootnet_backbone.emit(pb, t))
        gamma = pb.dense(P.depth_layer.emit(pb, feat))
        if gamma.C != 1:
            return ["x_root", "k_value"], [("dense", gamma, (N, gamma.C))], {"x_root": t}
        depth = pb.row_scale(gamma, kv)
        return ["x_root", "k_value"], [("dense", depth, (N, 1))], {"x_root": t}

    def forward(self, x_root, k_value):
        return self._run(x_root, k_value)[0]


def get_rootNetwithRegInt_model(init_params_dict, args, **kwargs):
    """Factory with the reference's checks (full_net.py:401-435) and DepthNet -> full transfer."""
    if args.backbone_name not in _RESNETS + _HRNETS:
        raise NotImplementedError
    if args.rootnet_backbone_name not in ["resnet", "resnet50", "resnet34"] + _HRNETS:
        raise NotImplementedError
    model = RootNetwithRegInt(init_params_dict, args, **kwargs)
    if args.pretrained_rootnet is not None:
        ckpt = torch.load(args.pretrained_rootnet, map_location="cpu")
        print(f"Using {args.pretrained_rootnet} as pretrained rootnet weights for rootNetwithRegInt pipeline. ")
        renamed = {(k.replace("backbone", "rootnet_backbone") if k.startswith("backbone") else k): v
                   for k, v in ckpt["model_state_dict"].items()}
        model.load_state_dict(renamed, strict=False)
    else:
        print("Not using pretrained depthnet weights for the full network training stage. ")
    return model
