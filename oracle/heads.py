"""Oracle: soft-argmax head, camera math, regressors, DepthNet / full-network forward, loss.

TEST INFRASTRUCTURE.  Restates (reference file:line):
  * HeatmapIntegralPose (hrnet branch)  - lib/utils/integral.py:97-105, 147-186
  * inverse intrinsics                  - lib/utils/transforms.py:145-162 (fp64 divide, fp32 store)
  * uvd_to_xyz / uvz2xyz_singlepoint    - lib/utils/transforms.py:33-73, 133-143
  * RootNet.forward                     - lib/models/depth_net.py:92-137
  * RootNetwithRegInt.forward           - lib/models/full_net.py:239-397
  * loss assembly (configs/panda/full.yaml) - lib/core/function.py:191-322
"""
import torch
import torch.nn.functional as F

from .fk import project
from .hrnet import hrnet_w32_forward


def inv_intrinsics(K):
    fx, fy = K[:, 0, 0].double(), K[:, 1, 1].double()
    cx, cy = K[:, 0, 2].double(), K[:, 1, 2].double()
    inv = torch.zeros(K.shape[0], 3, 3, dtype=torch.float32)
    inv[:, 0, 0] = 1.0 / fx
    inv[:, 0, 2] = -cx / fx
    inv[:, 1, 1] = 1.0 / fy
    inv[:, 1, 2] = -cy / fy
    inv[:, 2, 2] = 1
    return inv


def soft_argmax_uvd(out, num_joints=7, depth_dim=64, height=64, width=64, root=3, fix_root=True):
    """integral.py:147-177: softmax over D*H*W per joint, marginals, expectation/dim - 0.5."""
    B = out.shape[0]
    p = F.softmax(out.reshape(B, num_joints, -1), dim=2)
    p = p.reshape(B, num_joints, depth_dim, height, width)
    hx, hy, hz = p.sum((2, 3)), p.sum((2, 4)), p.sum((3, 4))
    r = torch.arange(hx.shape[-1], dtype=torch.float32).unsqueeze(-1)
    u = hx.matmul(r) / float(width) - 0.5
    v = hy.matmul(r) / float(height) - 0.5
    d = hz.matmul(r) / float(depth_dim) - 0.5
    uvd = torch.cat((u, v, d), dim=2)
    if fix_root:
        uvd = uvd.clone()
        uvd[:, root, 2] = 0.0
    return uvd


def uvd_to_xyz(uvd, K, z_root, image_size=256.0, depth_factor=1.3):
    """transforms.py:33-73 with return_relative=False; z_root is [B,1] (metres)."""
    uv1 = torch.cat(((uvd[:, :, :2] + 0.5) * image_size, torch.ones_like(uvd[:, :, 2:])), dim=2)
    xyz = torch.matmul(inv_intrinsics(K).unsqueeze(1), uv1.unsqueeze(-1)).squeeze(3)
    abs_z = uvd[:, :, 2] * depth_factor + z_root
    return xyz * abs_z.unsqueeze(-1)


def uvz2xyz_singlepoint(uv, z, K):
    """transforms.py:133-143."""
    v = torch.cat([uv * z, z], dim=1)
    return torch.matmul(inv_intrinsics(K), v.unsqueeze(-1)).squeeze(-1)


def rootnet_forward(sd, x, k_value, training=False, backbone="hrnet32", use_offset=False, add_fc=False, pred_xy=False):
    """RootNet(backbone).forward, depth_net.py:92-137 (pred_xy off); a ResNet trunk is followed by global average
    pooling (:93-95).  pred_xy (ResNet trunks only, as in the reference): three deconv + BN + ReLU layers and a 1x1
    conv on the feature map, softmax over the 64 x 64 map, expected column / row (:98-110); returns [x, y, depth].  add_fc: the residual MLP on the pooled feature (:113-120, four Linear + BatchNorm1d + ReLU and a
    fifth Linear added back); use_offset: depth += 1000 * offset_layer(feature) (:127-131)."""
    if backbone.startswith("resnet"):
        from .resnet import resnet_forward
        fm = resnet_forward(sd, x, prefix="backbone.", name="resnet50" if backbone == "resnet" else backbone, training=training)
        feat = fm.flatten(2).mean(2)
        if pred_xy:
            from .hrnet import _Ctx, _bn
            c, h = _Ctx(sd, "", training), fm
            for i in (0, 3, 6):
                h = F.relu(_bn(c, f"deconv_layers.{i + 1}", F.conv_transpose2d(h, sd[f"deconv_layers.{i}.weight"], None, stride=2, padding=1)))
            xy = F.conv2d(h, sd["xy_layer.weight"], sd["xy_layer.bias"])
            B, _, H, W = xy.shape
            p = F.softmax(xy.reshape(B, 1, H * W), 2).reshape(B, 1, H, W)
            coord_x = (p.sum(2) * torch.arange(W).float()).sum(2)
            coord_y = (p.sum(3) * torch.arange(H).float()).sum(2)
    else:
        feat = hrnet_w32_forward(sd, x, prefix="backbone.", generate_hm=False, generate_feat=True,
                                 training=training)
    if add_fc:
        h = feat
        for i in range(1, 5):
            h = F.linear(h, sd[f"depth_fc{i}.weight"], sd[f"depth_fc{i}.bias"])
            h = F.relu(F.batch_norm(h, sd[f"depth_bn{i}.running_mean"], sd[f"depth_bn{i}.running_var"], sd[f"depth_bn{i}.weight"],
                                    sd[f"depth_bn{i}.bias"], training, 0.1, 1e-5))
        feat = feat + F.linear(h, sd["depth_fc5.weight"], sd["depth_fc5.bias"])
    gamma = F.conv2d(feat[:, :, None, None], sd["depth_layer.weight"], sd["depth_layer.bias"])
    depth = gamma.view(-1, 1) * k_value.view(-1, 1)
    if use_offset:
        depth = depth + 1000.0 * F.conv2d(feat[:, :, None, None], sd["offset_layer.weight"], sd["offset_layer.bias"]).view(-1, 1)
    if pred_xy:
        return torch.cat((coord_x, coord_y, depth), dim=1)
    return depth


def _iter_reg(sd, xf, init, n_iter, fc1, fc2, dec):
    """full_net.py:318-331 / 365-378 with p_dropout = 0: p <- p + W3 (W2 (W1 [xf; p]))."""
    p = init
    for _ in range(n_iter):
        h = F.linear(torch.cat([xf, p], 1), sd[fc1 + ".weight"], sd[fc1 + ".bias"])
        h = F.linear(h, sd[fc2 + ".weight"], sd[fc2 + ".bias"])
        p = F.linear(h, sd[dec + ".weight"], sd[dec + ".bias"]) + p
    return p


def full_forward(sd, robot, x_reg, x_root, k_value, K, training=False, n_iter=4, root=3,
                 fix_root=True, image_size=256.0, depth_factor=1.3, reg_backbone="hrnet32", root_backbone="hrnet32",
                 direct_reg_rot=False, kps_need_depth=None, rot_iterative_matmul=False, add_fc=False, joint_bounds=None,
                 init_pose=None, init_rot=None):
    """RootNetwithRegInt.forward with rootnet_backbone_name = 'hrnet32' and backbone_name = 'hrnet32' or a ResNet
    with the deconv head (the shipped full.yaml) (full_net.py:239-397).  Returns the reference's 8-tuple."""
    B = x_reg.shape[0]
    init_pose = sd["init_pose"].expand(B, -1) if init_pose is None else init_pose     # :245-248
    init_rot = sd["init_rot"].expand(B, -1) if init_rot is None else init_rot
    if root_backbone.startswith("resnet"):                                            # :262-266
        from .resnet import resnet_forward
        feat_root = resnet_forward(sd, x_root, prefix="rootnet_backbone.",
                                   name="resnet50" if root_backbone == "resnet" else root_backbone, training=training).flatten(2).mean(2)
    else:
        feat_root = hrnet_w32_forward(sd, x_root, prefix="rootnet_backbone.", generate_hm=False,
                                      generate_feat=True, training=training)
    if add_fc:                                                                        # :261-270
        lin = lambda n, v: F.linear(v, sd[n + ".weight"], sd[n + ".bias"])   # noqa: E731
        f1 = lin("depth_fc_d1", feat_root)
        mid = F.leaky_relu(F.batch_norm(lin("depth_fc_d2", f1), sd["depth_bn.running_mean"], sd["depth_bn.running_var"],
                                        sd["depth_bn.weight"], sd["depth_bn.bias"], training, 0.1, 1e-5))
        f3 = 0.5 * (lin("depth_fc_u2", mid) + f1)
        feat_root = 0.5 * (lin("depth_fc_u1", f3) + feat_root)
    gamma = F.conv2d(feat_root[:, :, None, None], sd["depth_layer.weight"], sd["depth_layer.bias"])
    pred_depths = None
    if kps_need_depth is not None:                                                    # multi_kp, :275-279
        n = len(kps_need_depth)
        pred_depths = gamma.view(-1, n) * k_value.view(-1, 1).expand(-1, n) / 1000.0
        pred_depth = pred_depths[:, kps_need_depth.index(root)].reshape(-1, 1)
    else:
        pred_depth = (gamma.view(-1, 1) * k_value.view(-1, 1)).reshape(B, 1) / 1000.0   # :281-282
    if reg_backbone.startswith("resnet"):                                             # :293-296
        from .resnet import deconv_head_forward, resnet_forward
        x_out = resnet_forward(sd, x_reg, prefix="reg_backbone.", name="resnet50" if reg_backbone == "resnet" else reg_backbone,
                               training=training)
        heat, xf = deconv_head_forward(sd, x_out, training=training)
    else:
        heat, xf = hrnet_w32_forward(sd, x_reg, prefix="reg_backbone.", generate_hm=True,
                                     generate_feat=True, training=training)
    uvd = soft_argmax_uvd(heat, num_joints=len(robot.link_names), root=root, fix_root=fix_root)
    xyz_int = uvd_to_xyz(uvd, K, pred_depth, image_size, depth_factor)
    root_uv = (uvd[:, root, :2] + 0.5) * image_size                                  # :302
    trans = uvz2xyz_singlepoint(root_uv, pred_depth, K)                               # :305
    if joint_bounds is not None:    # reg_joint_map (full_net.py:313-316; HeatmapIntegralJoint, integral.py:206-232)
        from .hrnet import _Ctx, _bn
        c, j = _Ctx(sd, "", training), x_out
        for i in (0, 3, 6):
            j = F.relu(_bn(c, f"joint_conv_layers.{i + 1}", F.conv2d(j, sd[f"joint_conv_layers.{i}.weight"], sd[f"joint_conv_layers.{i}.bias"], padding=1)))
        jm = F.conv2d(j, sd["joint_final_layer.weight"], sd["joint_final_layer.bias"])
        hm = F.softmax(jm.reshape(B, jm.shape[1], -1), 2)
        coord = (hm * torch.arange(hm.shape[-1], dtype=torch.float32)).sum(2) / float(hm.shape[-1])
        jb = torch.as_tensor(joint_bounds, dtype=torch.float32)
        pose = coord * (jb[:, 1] - jb[:, 0]) + jb[:, 0]
    else:
        pose = _iter_reg(sd, xf, init_pose, n_iter, "fc_pose_1", "fc_pose_2", "decpose")
    if direct_reg_rot:      # full_net.py:333-345: six stacked Linear layers with one skip, no iteration, no init_rot
        lin = lambda n, v: F.linear(v, sd[n + ".weight"], sd[n + ".bias"])   # noqa: E731
        xc1 = lin("fc_rot_1", xf)
        xc = xc1
        for i in range(2, 7):
            xc = lin(f"fc_rot_{i}", xc)
        rot = lin("decrot", xc + xc1)
    elif rot_iterative_matmul:      # full_net.py:346-362: the decoded 6-vector is composed onto the estimate as a rotation
        from .fk import rot6d_to_rotmat, rotmat_to_rot6d
        rot = init_rot
        for _ in range(n_iter):
            h = F.linear(torch.cat([xf, rot], 1), sd["fc_rot_1.weight"], sd["fc_rot_1.bias"])
            h = F.linear(h, sd["fc_rot_2.weight"], sd["fc_rot_2.bias"])
            rot = rotmat_to_rot6d(rot6d_to_rotmat(F.linear(h, sd["decrot.weight"], sd["decrot.bias"])) @ rot6d_to_rotmat(rot))
    else:
        rot = _iter_reg(sd, xf, init_rot, n_iter, "fc_rot_1", "fc_rot_2", "decrot")
    xyz_fk = robot.get_keypoints_root(pose, rot, trans, root=root)                    # :380-383
    if pred_depths is not None:                                                       # :392-393
        return pose, rot, trans, root_uv, pred_depth, pred_depths, uvd, xyz_int, xyz_fk
    return pose, rot, trans, root_uv, pred_depth, uvd, xyz_int, xyz_fk


def full_loss(pred, gt, K, root=3, image_size=256.0):
    """function.py:191-322 under configs/panda/full.yaml:43-66 (weights 1,1,1(trans),10(depth),
    1(uv),10,10,10,10, align 0).  gt: dict with pose, root_rot, root_trans, root_uv, kp3d, kp2d,
    mask.  Returns (loss, dict of terms)."""
    pose, rot, trans, root_uv, depth, uvd, xyz_int, xyz_fk = pred
    uv_int = project(K, xyz_int)
    uv_fk = project(K, xyz_fk)
    m = gt["mask"]
    t = {}
    t["loss_joint"] = F.mse_loss(pose, gt["pose"])
    t["loss_rot"] = F.mse_loss(rot, gt["root_rot"])
    t["loss_depth"] = F.l1_loss(depth, gt["root_trans"][:, 2:3])
    e = torch.norm((root_uv - gt["root_uv"]) / image_size, dim=1) * m[:, root]
    t["loss_uv"] = e.sum() / (m[:, root] != 0).sum()
    e = torch.norm(trans - gt["root_trans"], dim=1)
    lt = e.mean()
    if lt > 0.5:                                                   # function.py:248-251
        lt = (e * torch.exp(-20.0 * e).detach()).mean()
    t["loss_trans"] = lt
    t["loss_error3d"] = torch.norm(xyz_fk - gt["kp3d"], dim=2).mean()
    gt2d = gt["kp2d"] / image_size
    t["loss_error2d"] = (torch.norm(uv_fk / image_size - gt2d, dim=2) * m).sum() / (m != 0).sum()
    t["loss_error3d_int"] = torch.norm(xyz_int - gt["kp3d"], dim=2).mean()
    t["loss_error2d_int"] = (torch.norm(uv_int / image_size - gt2d, dim=2) * m).sum() / (m != 0).sum()
    t["loss_error3d_align"] = torch.norm(xyz_fk - xyz_int, dim=2).mean()
    loss = (t["loss_joint"] + t["loss_rot"] + t["loss_uv"] + 10.0 * t["loss_depth"] + t["loss_trans"]
            + 10.0 * t["loss_error2d"] + 10.0 * t["loss_error3d"] + 10.0 * t["loss_error2d_int"]
            + 10.0 * t["loss_error3d_int"] + 0.0 * t["loss_error3d_align"])
    return loss, t
