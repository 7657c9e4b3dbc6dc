"""CPU: the oracle (oracle/*.py) against every golden fixture produced by the reference itself."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, PANDA_URDF
from oracle import fk, heads, hrnet
from synth import synth_inputs, synth_state_dict

torch.set_num_threads(8)


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def check_summary(t, g, key, rtol, atol):
    f = t.detach().reshape(-1).double()
    s = g[key + "summary"] if (key + "summary") in g else g[key + "_summary"]
    idx = g[key + "idx"] if (key + "idx") in g else g[key + "_idx"]
    val = g[key + "val"] if (key + "val") in g else g[key + "_val"]
    np.testing.assert_allclose(f.abs().mean().item(), s[1], rtol=rtol)
    np.testing.assert_allclose(f[idx].float().numpy(), val, rtol=rtol, atol=atol)


def hrnet_shapes(prefix="", hm=True, feat=True, depth_dim=64, num_joints=7):
    """State-dict key -> shape of the reference HRNet-W32 (built from the product module tree)."""
    from hrpe_amd.lib.models.backbones.HRnet import get_hrnet
    m = get_hrnet(32, num_joints, depth_dim, pretrain=False, generate_feat=feat, generate_hm=hm)
    return {prefix + k: v for k, v in m.state_dict().items()}


@pytest.fixture(scope="module")
def robot():
    return fk.Robot(PANDA_URDF)


def test_fk_known_answer_limb_lengths(robot):
    """Weights-free known answer: FK at q=0 reproduces PANDA_LIMB_LENGTH (reference const.py:100-107)."""
    p = robot.get_keypoints_only_fk(torch.zeros(1, 8))[0]
    d = torch.norm(p[1:] - p[:-1], dim=1).numpy()
    np.testing.assert_allclose(d, [0.3330, 0.3160, 0.0825, 0.39276, 0.0880, 0.1070], atol=2e-5)


def test_fk_golden(robot):
    g = load("golden_fk.npz")
    q, r, t, K = [torch.tensor(g[k]) for k in ("q", "rot6d", "t", "K")]
    np.testing.assert_allclose(robot.get_keypoints_only_fk(q).numpy(), g["fk_only"], atol=1e-6)
    np.testing.assert_allclose(robot.get_keypoints_only_fk(torch.zeros(1, 8)).numpy(), g["fk_q0"], atol=1e-7)
    for root in (0, 3):
        tq, tr, tt = [x.clone().requires_grad_(True) for x in (q, r, t)]
        xyz = robot.get_keypoints_root(tq, tr, tt, root=root)
        uv = fk.project(K, xyz)
        np.testing.assert_allclose(xyz.detach().numpy(), g[f"xyz_root{root}"], atol=2e-6)
        np.testing.assert_allclose(uv.detach().numpy(), g[f"uv_root{root}"], atol=2e-3, rtol=1e-5)
        ((xyz * torch.tensor(g["w_xyz"])).sum() + (uv * torch.tensor(g["w_uv"])).sum()).backward()
        for name, x in (("gq", tq), ("grot", tr), ("gt", tt)):
            ref = g[f"{name}_root{root}"]
            np.testing.assert_allclose(x.grad.numpy(), ref, atol=2e-4 * max(1.0, np.abs(ref).max()), rtol=1e-3)
        rr = robot.get_rotation_at_specific_root(q, r, t, root=root)
        np.testing.assert_allclose(rr.numpy(), g[f"rootrot_root{root}"], atol=2e-6)


_OTHER = {"kuka": "kuka_kinematics.urdf", "baxter": "baxter_kinematics.urdf"}


def test_fk_known_answer_kuka_limb_lengths():
    """FK at q=0 reproduces KUKA_LIMB_LENGTH (reference const.py:108-116)."""
    rb = fk.Robot(os.path.join(os.path.dirname(PANDA_URDF), _OTHER["kuka"]), "kuka")
    p = rb.get_keypoints_only_fk(torch.zeros(1, 7))[0]
    np.testing.assert_allclose(torch.norm(p[1:] - p[:-1], dim=1).numpy(),
                               [0.15, 0.19, 0.21, 0.19, 0.21, 0.19946, 0.10122], atol=2e-5)


@pytest.mark.parametrize("robot_type", ["kuka", "baxter"])
def test_fk_golden_other_robots(robot_type):
    """The 7-DoF serial chain and the 15-DoF tree with keypoint offsets (urdf_robot.py:57-74) against
    fixtures produced by the reference's URDFRobot on the same URDF files."""
    rb = fk.Robot(os.path.join(os.path.dirname(PANDA_URDF), _OTHER[robot_type]), robot_type)
    g = load(f"golden_fk_{robot_type}.npz")
    q, r, t, K = [torch.tensor(g[k]) for k in ("q", "rot6d", "t", "K")]
    assert q.shape[1] == rb.dof and g["fk_only"].shape[1] == len(rb.link_names)
    np.testing.assert_allclose(rb.get_keypoints_only_fk(q).numpy(), g["fk_only"], atol=1e-6)
    np.testing.assert_allclose(rb.get_keypoints_only_fk(torch.zeros(1, rb.dof)).numpy(), g["fk_q0"], atol=1e-7)
    for root in g["roots"].tolist():
        tq, tr, tt = [x.clone().requires_grad_(True) for x in (q, r, t)]
        xyz = rb.get_keypoints_root(tq, tr, tt, root=root)
        uv = fk.project(K, xyz)
        np.testing.assert_allclose(xyz.detach().numpy(), g[f"xyz_root{root}"], atol=3e-6)
        np.testing.assert_allclose(uv.detach().numpy(), g[f"uv_root{root}"], atol=3e-3, rtol=1e-5)
        ((xyz * torch.tensor(g["w_xyz"])).sum() + (uv * torch.tensor(g["w_uv"])).sum()).backward()
        for name, x in (("gq", tq), ("grot", tr), ("gt", tt)):
            ref = g[f"{name}_root{root}"]
            np.testing.assert_allclose(x.grad.numpy(), ref, atol=2e-4 * max(1.0, np.abs(ref).max()), rtol=1e-3)
        rr = rb.get_rotation_at_specific_root(q, r, t, root=root)
        np.testing.assert_allclose(rr.numpy(), g[f"rootrot_root{root}"], atol=2e-6)


def test_integral_golden():
    g = load("golden_integral.npz")
    rng = np.random.Generator(np.random.PCG64(int(g["seed"])))
    out = torch.as_tensor(rng.normal(0, 2.0, (2, 7 * 64, 64, 64)).astype(np.float32))
    out[:, ::5] += 3.0
    out.requires_grad_(True)
    uvd = heads.soft_argmax_uvd(out)
    xyz = heads.uvd_to_xyz(uvd, torch.tensor(g["K"]), torch.tensor(g["z_root"]))
    np.testing.assert_allclose(uvd.detach().numpy(), g["uvd"], atol=1e-6)
    np.testing.assert_allclose(xyz.detach().numpy(), g["xyz"], atol=1e-6)
    (uvd * torch.tensor(g["w"])).sum().backward()
    check_summary(out.grad, g, "g_", rtol=1e-4, atol=1e-12)


def test_hrnet_eval_golden():
    g = load("golden_hrnet_eval.npz")
    sd = synth_state_dict(hrnet_shapes())
    x, _, _, _ = synth_inputs(2)
    taps = {}
    with torch.no_grad():
        heat, feat = hrnet.hrnet_w32_forward(sd, x, taps=taps)
    np.testing.assert_allclose(feat.numpy(), g["feat"], atol=1e-6)
    check_summary(heat, g, "heat_", rtol=1e-6, atol=1e-6)
    flat = {"stem": taps["stem"], "layer1": taps["layer1"], "head_map": taps["head_map"]}
    for st in ("stage2", "stage3", "stage4"):
        for b, t in enumerate(taps[st]):
            flat[f"{st}_{b}"] = t
    for k, t in flat.items():
        check_summary(t, g, f"tap_{k}_", rtol=1e-6, atol=1e-6)


def depthnet_sd():
    shapes = hrnet_shapes("backbone.", hm=False, feat=True)
    shapes["depth_layer.weight"] = torch.empty(1, 2048, 1, 1)
    shapes["depth_layer.bias"] = torch.empty(1)
    return synth_state_dict(shapes)


def test_depthnet_golden():
    g = load("golden_depthnet.npz")
    sd = depthnet_sd()
    x, _, kv, _ = synth_inputs(2)
    with torch.no_grad():
        d = heads.rootnet_forward(sd, x, kv)
    np.testing.assert_allclose(d.numpy(), g["depth_eval"], rtol=1e-6)
    for k, v in sd.items():
        if v.dtype.is_floating_point and "running" not in k:
            v.requires_grad_(True)
    pred = heads.rootnet_forward(sd, x, kv, training=True) / 1000.0
    loss = torch.nn.functional.l1_loss(pred, torch.tensor(g["gt_depth"]))
    loss.backward()
    np.testing.assert_allclose(pred.detach().numpy(), g["depth_train"], rtol=1e-5)
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=1e-5)
    for key in g.files:
        if key.startswith("grad:") and key.endswith(":val"):
            name = key.split(":")[1]
            check_summary(sd[name].grad, g, f"grad:{name}:", rtol=2e-3, atol=1e-9)
        if key.startswith("buf:"):
            np.testing.assert_allclose(sd[key[4:]][:64].detach().numpy(), g[key], rtol=1e-5, atol=1e-7)


def depthnet_variants_sd():
    shapes = hrnet_shapes("backbone.", hm=False, feat=True)
    for i, (o, c) in enumerate([(1024, 2048), (512, 1024), (512, 512), (1024, 512), (2048, 1024)], 1):
        shapes[f"depth_fc{i}.weight"], shapes[f"depth_fc{i}.bias"] = torch.empty(o, c), torch.empty(o)
        if i < 5:
            for leaf in ("weight", "bias", "running_mean", "running_var"):
                shapes[f"depth_bn{i}.{leaf}"] = torch.empty(o)
            shapes[f"depth_bn{i}.num_batches_tracked"] = torch.zeros((), dtype=torch.long)
    for n in ("depth_layer", "offset_layer"):
        shapes[n + ".weight"], shapes[n + ".bias"] = torch.empty(1, 2048, 1, 1), torch.empty(1)
    return synth_state_dict(shapes)


def test_depthnet_variants_golden():
    """RootNet(use_offset=True, add_fc=True) (depth_net.py:44-70, 113-131): oracle against the reference's outputs."""
    g = load("golden_depthnet_variants.npz")
    sd = depthnet_variants_sd()
    x, _, kv, _ = synth_inputs(8)
    with torch.no_grad():
        d = heads.rootnet_forward(sd, x, kv, use_offset=True, add_fc=True)
    np.testing.assert_allclose(d.numpy(), g["depth_eval"], rtol=1e-6)
    for k, v in sd.items():
        if v.dtype.is_floating_point and "running" not in k:
            v.requires_grad_(True)
    pred = heads.rootnet_forward(sd, x, kv, training=True, use_offset=True, add_fc=True) / 1000.0
    loss = torch.nn.functional.l1_loss(pred, torch.tensor(g["gt_depth"]))
    loss.backward()
    np.testing.assert_allclose(pred.detach().numpy(), g["depth_train"], rtol=1e-5)
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=1e-5)
    for key in g.files:
        if key.startswith("grad:") and key.endswith(":val"):
            name = key.split(":")[1]
            check_summary(sd[name].grad, g, f"grad:{name}:", rtol=2e-3, atol=1e-9)
        if key.startswith("buf:"):
            np.testing.assert_allclose(sd[key[4:]][:64].detach().numpy(), g[key], rtol=1e-5, atol=1e-7)


def full_sd(dof=8, nkp=7, init_pose=(0.0, 0.0, 0.0, -1.52715, 0.0, 1.8675, 0.0, 0.02)):
    shapes = {}
    shapes.update(hrnet_shapes("reg_backbone.", hm=True, feat=True, num_joints=nkp))
    shapes.update(hrnet_shapes("rootnet_backbone.", hm=False, feat=True))
    for n, (o, i) in {"fc_pose_1": (1024, 2048 + dof), "fc_pose_2": (1024, 1024), "decpose": (dof, 1024),
                      "fc_rot_1": (1024, 2054), "fc_rot_2": (1024, 1024), "decrot": (6, 1024)}.items():
        shapes[n + ".weight"] = torch.empty(o, i)
        shapes[n + ".bias"] = torch.empty(o)
    shapes["depth_layer.weight"] = torch.empty(1, 2048, 1, 1)
    shapes["depth_layer.bias"] = torch.empty(1)
    # reference const.py:168-178 mean pose / identity camera rotation (full_net.py:179-192)
    shapes["init_pose"] = torch.tensor([list(init_pose)])
    shapes["init_rot"] = torch.tensor([[1.0, 0.0, 0.0, 0.0, 1.0, 0.0]])
    return synth_state_dict(shapes)


NAMES8 = ["pose", "rot", "trans", "root_uv", "depth", "uvd", "xyz_int", "xyz_fk"]


def full_sd_resnet():
    """Shipped full.yaml: ResNet-50 regression trunk + deconv head, HRNet-W32 root trunk (shapes from the product
    module tree, whose keys tests/test_host_cpu.py pins)."""
    from hrpe_amd.lib.models.backbones.Resnet import get_resnet
    sd = full_sd()
    sd = {k: v for k, v in sd.items() if not k.startswith("reg_backbone.")}
    shapes = {"reg_backbone." + k: v for k, v in get_resnet("resnet50", pretrain=False).state_dict().items()}
    cin = 2048
    for i in (0, 3, 6):
        shapes[f"deconv_layers.{i}.weight"] = torch.empty(cin, 256, 4, 4)
        for n, shp in (("weight", 256), ("bias", 256), ("running_mean", 256), ("running_var", 256)):
            shapes[f"deconv_layers.{i + 1}.{n}"] = torch.empty(shp)
        shapes[f"deconv_layers.{i + 1}.num_batches_tracked"] = torch.tensor(0, dtype=torch.long)
        cin = 256
    shapes["final_layer.weight"] = torch.empty(448, 256, 1, 1)
    shapes["final_layer.bias"] = torch.empty(448)
    sd.update(synth_state_dict(shapes))
    return sd


def test_full_eval_resnet_golden(robot):
    """backbone_name = 'resnet50' (Resnet.py:56-67 + full_net.py:194-216, 293-298) against the reference."""
    g = load("golden_full_eval_resnet.npz")
    sd = full_sd_resnet()
    x_reg, x_root, kv, K = synth_inputs(2)
    from oracle import resnet as ores
    with torch.no_grad():
        x_out = ores.resnet_forward(sd, x_reg, prefix="reg_backbone.")
        heat, _ = ores.deconv_head_forward(sd, x_out)
        out = heads.full_forward(sd, robot, x_reg, x_root, kv, K, reg_backbone="resnet50")
    np.testing.assert_allclose(x_out[:, ::64].numpy(), g["tap:x_out"], atol=1e-5, rtol=1e-5)
    np.testing.assert_allclose(heat[:, ::56, ::4, ::4].numpy(), g["tap:heat"], atol=1e-5, rtol=1e-5)
    for n, t in zip(NAMES8, out):
        np.testing.assert_allclose(t.numpy(), g[n], atol=1e-5, rtol=1e-5, err_msg=n)


def test_full_train_resnet_golden(robot):
    """One reference training step with the ResNet-50 regression trunk (function.py farward_loss, train=True)."""
    g = load("golden_full_train_resnet.npz")
    sd = full_sd_resnet()
    for k, v in sd.items():
        if v.dtype.is_floating_point and "running" not in k and not k.startswith("init_"):
            v.requires_grad_(True)
    rng = np.random.Generator(np.random.PCG64(2024))
    x_reg = torch.tensor(rng.integers(0, 256, (2, 3, 256, 256)).astype(np.float32)) / 255.
    x_root = torch.tensor(rng.integers(0, 256, (2, 3, 256, 256)).astype(np.float32)) / 255.
    K, kv = torch.tensor(g["in:K"]), torch.tensor(g["k_values"])
    q, R, t = torch.tensor(g["in:q"]), torch.tensor(g["in:R"]), torch.tensor(g["in:t"])
    kp3d, kp2d, mask = torch.tensor(g["in:kp3d"]), torch.tensor(g["in:kp2d"]), torch.tensor(g["in:mask"])
    gt = dict(pose=q, root_rot=robot.get_rotation_at_specific_root(q, fk.rotmat_to_rot6d(R), t, root=3),
              root_trans=kp3d[:, 3], root_uv=kp2d[:, 3], kp3d=kp3d, kp2d=kp2d, mask=mask)
    pred = heads.full_forward(sd, robot, x_reg, x_root, kv, K, training=True, reg_backbone="resnet50")
    for n, p in zip(NAMES8, pred):
        np.testing.assert_allclose(p.detach().numpy(), g["fwd:" + n], rtol=2e-4, atol=2e-5, err_msg=n)
    loss, terms = heads.full_loss(pred, gt, K)
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=1e-5)
    loss.backward()
    for key in g.files:
        if key.startswith("grad:") and key.endswith(":val"):
            name = key.split(":")[1]
            # atol: fp32 summation order of the 256x256 stem gradient differs between runs (values up to 0.75)
            check_summary(sd[name].grad, g, f"grad:{name}:", rtol=5e-3, atol=1e-5)
        if key.startswith("buf:"):
            np.testing.assert_allclose(sd[key[4:]][:64].detach().numpy(), g[key], rtol=1e-5, atol=1e-7)


def test_full_eval_golden(robot):
    g = load("golden_full_eval.npz")
    sd = full_sd()
    x_reg, x_root, kv, K = synth_inputs(2)
    with torch.no_grad():
        out = heads.full_forward(sd, robot, x_reg, x_root, kv, K)
    for n, t in zip(NAMES8, out):
        np.testing.assert_allclose(t.numpy(), g[n], atol=1e-5, rtol=1e-5, err_msg=n)


def test_full_eval_per_call_init_golden(robot):
    """forward(..., init_pose=, init_rot=) of the reference (full_net.py:239, 245-248; fixture written by the imported reference)."""
    g = load("golden_full_eval_init.npz")
    sd = full_sd()
    x_reg, x_root, kv, K = synth_inputs(2)
    with torch.no_grad():
        out = heads.full_forward(sd, robot, x_reg, x_root, kv, K, init_pose=torch.tensor(g["init_pose"]), init_rot=torch.tensor(g["init_rot"]))
    for n, t in zip(NAMES8, out):
        np.testing.assert_allclose(t.numpy(), g[n], atol=1e-5, rtol=1e-5, err_msg=n)
    assert np.abs(g["pose"] - load("golden_full_eval.npz")["pose"]).max() > 1e-3      # (the start matters)


def test_full_eval_direct_rot_golden(robot):
    """direct_reg_rot = True (full_net.py:105-127, 333-345)."""
    g = load("golden_full_eval_direct_rot.npz")
    sd = {k: v for k, v in full_sd().items() if not k.startswith(("fc_rot_", "decrot"))}
    shapes = {"fc_rot_1.weight": torch.empty(1024, 2048), "fc_rot_1.bias": torch.empty(1024),
              "decrot.weight": torch.empty(6, 1024), "decrot.bias": torch.empty(6)}
    for i in range(2, 7):
        shapes[f"fc_rot_{i}.weight"], shapes[f"fc_rot_{i}.bias"] = torch.empty(1024, 1024), torch.empty(1024)
    sd.update(synth_state_dict(shapes))
    x_reg, x_root, kv, K = synth_inputs(2)
    with torch.no_grad():
        out = heads.full_forward(sd, robot, x_reg, x_root, kv, K, direct_reg_rot=True)
    for n, t in zip(NAMES8, out):
        np.testing.assert_allclose(t.numpy(), g[n], atol=1e-5, rtol=1e-5, err_msg=n)


def test_full_eval_multi_kp_golden(robot):
    """multi_kp = True, kps_need_depth = [0, 3, 6] (full_net.py:146-148, 275-279, 392-393): the 9-tuple."""
    g = load("golden_full_eval_multi_kp.npz")
    sd = {k: v for k, v in full_sd().items() if not k.startswith("depth_layer")}
    sd.update(synth_state_dict({"depth_layer.weight": torch.empty(3, 2048, 1, 1), "depth_layer.bias": torch.empty(3)}))
    x_reg, x_root, kv, K = synth_inputs(2)
    with torch.no_grad():
        out = heads.full_forward(sd, robot, x_reg, x_root, kv, K, kps_need_depth=[0, 3, 6])
    assert len(out) == 9
    for n, t in zip(NAMES8[:5] + ["depths"] + NAMES8[5:], out):
        np.testing.assert_allclose(t.numpy(), g[n], atol=1e-5, rtol=1e-5, err_msg=n)


def test_full_eval_rot_matmul_golden(robot):
    """rot_iterative_matmul = True (full_net.py:346-362)."""
    g = load("golden_full_eval_rot_matmul.npz")
    sd = full_sd()
    x_reg, x_root, kv, K = synth_inputs(2)
    with torch.no_grad():
        out = heads.full_forward(sd, robot, x_reg, x_root, kv, K, rot_iterative_matmul=True)
    for n, t in zip(NAMES8, out):
        np.testing.assert_allclose(t.numpy(), g[n], atol=1e-5, rtol=1e-5, err_msg=n)


def _add_fc_sd():
    sd = full_sd()
    shapes = {}
    for n, (o, i) in {"depth_fc_d1": (1024, 2048), "depth_fc_d2": (512, 1024), "depth_fc_u2": (1024, 512), "depth_fc_u1": (2048, 1024)}.items():
        shapes[n + ".weight"], shapes[n + ".bias"] = torch.empty(o, i), torch.empty(o)
    for leaf in ("weight", "bias", "running_mean", "running_var"):
        shapes["depth_bn." + leaf] = torch.empty(512)
    shapes["depth_bn.num_batches_tracked"] = torch.zeros((), dtype=torch.long)
    sd.update(synth_state_dict(shapes))
    return sd


def test_full_add_fc_golden(robot):
    """add_fc = True (full_net.py:150-157, 261-270): eval at B = 2, a train-mode forward at B = 8."""
    g = load("golden_full_add_fc.npz")
    sd = _add_fc_sd()
    x_reg, x_root, kv, K = synth_inputs(2)
    with torch.no_grad():
        out = heads.full_forward(sd, robot, x_reg, x_root, kv, K, add_fc=True)
    for n, t in zip(NAMES8, out):
        np.testing.assert_allclose(t.numpy(), g["eval:" + n], atol=1e-5, rtol=1e-5, err_msg=n)
    x_reg, x_root, kv, K = synth_inputs(8)
    with torch.no_grad():
        out = heads.full_forward(sd, robot, x_reg, x_root, kv, K, training=True, add_fc=True)
    np.testing.assert_allclose(out[4].numpy(), g["train:depth"], rtol=2e-4)
    np.testing.assert_allclose(sd["depth_bn.running_mean"][:64].numpy(), g["buf:depth_bn.running_mean"], rtol=1e-4, atol=1e-6)


def test_depthnet_pred_xy_golden():
    """RootNet('resnet50', pred_xy=True) (depth_net.py:33-43, 98-110, 133-135)."""
    from hrpe_amd.lib.models.backbones.Resnet import get_resnet
    g = load("golden_depthnet_pred_xy.npz")
    shapes = {"backbone." + k: v for k, v in get_resnet("resnet50", pretrain=False).state_dict().items()}
    cin = 2048
    for i in (0, 3, 6):
        shapes[f"deconv_layers.{i}.weight"] = torch.empty(cin, 256, 4, 4)
        for leaf in ("weight", "bias", "running_mean", "running_var"):
            shapes[f"deconv_layers.{i + 1}.{leaf}"] = torch.empty(256)
        shapes[f"deconv_layers.{i + 1}.num_batches_tracked"] = torch.zeros((), dtype=torch.long)
        cin = 256
    shapes["xy_layer.weight"], shapes["xy_layer.bias"] = torch.empty(1, 256, 1, 1), torch.empty(1)
    shapes["depth_layer.weight"], shapes["depth_layer.bias"] = torch.empty(1, 2048, 1, 1), torch.empty(1)
    sd = synth_state_dict(shapes)
    x, _, kv, _ = synth_inputs(4)
    with torch.no_grad():
        d = heads.rootnet_forward(sd, x, kv, backbone="resnet50", pred_xy=True)
    np.testing.assert_allclose(d.numpy(), g["coord_eval"], rtol=1e-5)
    for k, v in sd.items():
        if v.dtype.is_floating_point and "running" not in k:
            v.requires_grad_(True)
    pred = heads.rootnet_forward(sd, x, kv, training=True, backbone="resnet50", pred_xy=True)
    loss = (pred[:, :2] / 64.0).square().sum() + (pred[:, 2:] / 1000.0).square().sum()
    loss.backward()
    np.testing.assert_allclose(pred.detach().numpy(), g["coord_train"], rtol=1e-5)
    for key in g.files:
        if key.startswith("grad:") and key.endswith(":val"):
            name = key.split(":")[1]
            check_summary(sd[name].grad, g, f"grad:{name}:", rtol=5e-3, atol=1e-8)


def _joint_map_sd():
    sd = {k: v for k, v in full_sd_resnet().items() if not k.startswith(("fc_pose_", "decpose"))}
    shapes, cin = {}, 2048
    for i in (0, 3, 6):
        shapes[f"joint_conv_layers.{i}.weight"], shapes[f"joint_conv_layers.{i}.bias"] = torch.empty(128, cin, 3, 3), torch.empty(128)
        for leaf in ("weight", "bias", "running_mean", "running_var"):
            shapes[f"joint_conv_layers.{i + 1}.{leaf}"] = torch.empty(128)
        shapes[f"joint_conv_layers.{i + 1}.num_batches_tracked"] = torch.zeros((), dtype=torch.long)
        cin = 128
    shapes["joint_final_layer.weight"], shapes["joint_final_layer.bias"] = torch.empty(8, 128, 1, 1), torch.empty(8)
    sd.update(synth_state_dict(shapes))
    return sd


def test_full_eval_joint_map_golden(robot):
    """reg_joint_map = True with a ResNet-50 regression trunk (full_net.py:87-93, 218-237, 313-316; integral.py:186-232)."""
    from hrpe_amd.lib.dataset.const import JOINT_BOUNDS
    g = load("golden_full_eval_joint_map.npz")
    sd = _joint_map_sd()
    x_reg, x_root, kv, K = synth_inputs(2)
    with torch.no_grad():
        out = heads.full_forward(sd, robot, x_reg, x_root, kv, K, reg_backbone="resnet50", joint_bounds=JOINT_BOUNDS["panda"])
    for n, t in zip(NAMES8, out):
        np.testing.assert_allclose(t.numpy(), g[n], atol=1e-5, rtol=1e-5, err_msg=n)


def test_full_eval_baxter_golden():
    """robot_type = 'baxter' (full_net.py:48-50): 15 DoF, 17 key-points -> 1088 heat-map channels, tree FK with
    key-point offsets; init pose = const.py:183-199 mean."""
    g = load("golden_full_eval_baxter.npz")
    rb = fk.Robot(os.path.join(os.path.dirname(PANDA_URDF), _OTHER["baxter"]), "baxter")
    mean = [0.0, 0.0, 0.0, -0.5499999999999999, -0.5499999999999999, 0.0, 0.0, 1.284, 1.284, 0.0, 0.0,
            0.2616018366049999, 0.2616018366049999, 0.0, 0.0]
    sd = full_sd(dof=15, nkp=17, init_pose=mean)
    x_reg, x_root, kv, K = synth_inputs(2)
    with torch.no_grad():
        out = heads.full_forward(sd, rb, x_reg, x_root, kv, K)
    for n, t in zip(NAMES8, out):
        assert tuple(t.shape) == g[n].shape, n
        np.testing.assert_allclose(t.numpy(), g[n], atol=1e-5, rtol=1e-5, err_msg=n)


@pytest.mark.parametrize("frozen_bn", [False, True, "quat"], ids=["train", "bn_eval", "quat"])
def test_full_train_golden(robot, frozen_bn):
    """One reference training step (lib/core/function.py farward_loss, train=True): loss terms,
    gradients and BN running stats.  bn_eval: the same step with every BatchNorm module in eval() as
    scripts/train_sim2real.py:139-146 trains (BASELINE config 5): running statistics, gradients through them."""
    quat = frozen_bn == "quat"            # rotation_dim = 4 (full_net.py:186-189, function.py:63-64): same step, quaternion rotations
    frozen_bn = frozen_bn is True
    g = load("golden_full_train_quat.npz" if quat else "golden_full_train_bn_eval.npz" if frozen_bn else "golden_full_train.npz")
    sd = _full_sd_quat() if quat else full_sd()
    for k, v in sd.items():
        if v.dtype.is_floating_point and "running" not in k and not k.startswith("init_"):
            v.requires_grad_(True)
    rng = np.random.Generator(np.random.PCG64(2024))
    x_reg = torch.tensor(rng.integers(0, 256, (2, 3, 256, 256)).astype(np.float32)) / 255.
    x_root = torch.tensor(rng.integers(0, 256, (2, 3, 256, 256)).astype(np.float32)) / 255.
    K = torch.tensor(g["in:K"])
    kv = torch.tensor(g["k_values"])
    q, R, t = torch.tensor(g["in:q"]), torch.tensor(g["in:R"]), torch.tensor(g["in:t"])
    kp3d, kp2d, mask = torch.tensor(g["in:kp3d"]), torch.tensor(g["in:kp2d"]), torch.tensor(g["in:mask"])
    gt = dict(pose=q, root_rot=robot.get_rotation_at_specific_root(q, (fk.rotmat_to_quat if quat else fk.rotmat_to_rot6d)(R), t, root=3),
              root_trans=kp3d[:, 3], root_uv=kp2d[:, 3], kp3d=kp3d, kp2d=kp2d, mask=mask)
    pred = heads.full_forward(sd, robot, x_reg, x_root, kv, K, training=not frozen_bn)
    for n, p in zip(NAMES8, pred):
        np.testing.assert_allclose(p.detach().numpy(), g["fwd:" + n], rtol=2e-4, atol=2e-5, err_msg=n)
    loss, terms = heads.full_loss(pred, gt, K)
    for k, v in terms.items():
        np.testing.assert_allclose(v.item(), g["term:" + k], rtol=1e-4, err_msg=k)
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=1e-5)
    loss.backward()
    for key in g.files:
        if key.startswith("grad:") and key.endswith(":val"):
            name = key.split(":")[1]
            check_summary(sd[name].grad, g, f"grad:{name}:", rtol=5e-3, atol=1e-8)
        if key.startswith("buf:"):
            np.testing.assert_allclose(sd[key[4:]][:64].detach().numpy(), g[key], rtol=1e-5, atol=1e-7)


def test_depthnet_resnet_golden(robot):
    """ResNet-50 as the DepthNet trunk and as both trunks of the full network (depth_net.py:16-18, 93-95;
    full_net.py:136-138, 262-266)."""
    from hrpe_amd.lib.models.backbones.Resnet import get_resnet
    g = load("golden_depthnet_resnet.npz")
    shapes = {"backbone." + k: v for k, v in get_resnet("resnet50", pretrain=False).state_dict().items()}
    shapes["depth_layer.weight"], shapes["depth_layer.bias"] = torch.empty(1, 2048, 1, 1), torch.empty(1)
    sd = synth_state_dict(shapes)
    x, _, kv, _ = synth_inputs(2)
    with torch.no_grad():
        d = heads.rootnet_forward(sd, x, kv, backbone="resnet50")
    np.testing.assert_allclose(d.numpy(), g["depth_eval"], rtol=1e-5)
    for k, v in sd.items():
        if v.dtype.is_floating_point and "running" not in k:
            v.requires_grad_(True)
    pred = heads.rootnet_forward(sd, x, kv, training=True, backbone="resnet50") / 1000.0
    loss = torch.nn.functional.l1_loss(pred, torch.tensor(g["gt_depth"]))
    loss.backward()
    np.testing.assert_allclose(pred.detach().numpy(), g["depth_train"], rtol=1e-5)
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=1e-5)
    for key in g.files:
        if key.startswith("grad:") and key.endswith(":val"):
            name = key.split(":")[1]
            check_summary(sd[name].grad, g, f"grad:{name}:", rtol=5e-3, atol=1e-8)
    fsd = full_sd_resnet()
    fsd = {k: v for k, v in fsd.items() if not k.startswith("rootnet_backbone.")}
    fsd.update(synth_state_dict({"rootnet_backbone." + k: v for k, v in get_resnet("resnet50", pretrain=False).state_dict().items()}))
    x_reg, x_root, kv, K = synth_inputs(2)
    with torch.no_grad():
        out = heads.full_forward(fsd, robot, x_reg, x_root, kv, K, reg_backbone="resnet50", root_backbone="resnet50")
    for n, t in zip(NAMES8, out):
        np.testing.assert_allclose(t.numpy(), g["full:" + n], atol=1e-5, rtol=1e-5, err_msg=n)


def _two_iterations(sd, loss_fn, clip, g, rtol_loss):
    """The trainers' loop twice on the oracle's state dict (scripts/train_full.py:56-66): zero_grad, loss, backward,
    clip_grad_norm_, Adam(lr 1e-4).step, checked against the reference's own two iterations."""
    params = [v for k, v in sd.items() if v.requires_grad]
    p0 = {k: v.detach().clone() for k, v in sd.items() if v.requires_grad}
    opt = torch.optim.Adam(params, lr=1e-4, weight_decay=0.0)
    for it in range(2):
        opt.zero_grad()
        loss = loss_fn()
        loss.backward()
        norm = torch.nn.utils.clip_grad_norm_(params, clip)
        opt.step()
        np.testing.assert_allclose(loss.item(), g[f"loss{it + 1}"], rtol=rtol_loss, err_msg=f"loss of iteration {it + 1}")
        np.testing.assert_allclose(float(norm), g[f"grad_norm{it + 1}"], rtol=5e-3, err_msg=f"gradient norm of iteration {it + 1}")
    for key in g.files:
        if key.startswith("upd:") and key.endswith(":val"):
            name = key.split(":")[1]
            upd = (sd[name].detach() - p0[name]).reshape(-1)[g[f"upd:{name}:idx"]].numpy()
            ref = g[key]
            # Adam's first steps are ~ lr * sign(g): an element whose gradient is rounding noise may land on the other side
            err = np.abs(upd - ref)
            assert np.median(err) < 0.02 * g[f"upd:{name}:absmean"] and np.mean(err > 0.5 * g[f"upd:{name}:absmean"]) < 0.03, \
                (name, np.median(err), g[f"upd:{name}:absmean"], np.mean(err > 0.5 * g[f"upd:{name}:absmean"]))
        if key.startswith("buf:") and "num_batches" not in key:
            np.testing.assert_allclose(sd[key[4:]].reshape(-1)[:64].detach().numpy(), g[key], rtol=1e-4, atol=1e-6)


def test_depthnet_two_iterations_golden():
    """BASELINE.json configs[0]: depthnet.yaml, B = 4, two iterations of the DepthNet trainer (golden_depthnet_2iter.npz)."""
    g = load("golden_depthnet_2iter.npz")
    sd = depthnet_sd()
    for k, v in sd.items():
        if v.dtype.is_floating_point and "running" not in k:
            v.requires_grad_(True)
    x, _, kv, _ = synth_inputs(4)
    gt = torch.tensor(g["gt_depth"])
    _two_iterations(sd, lambda: torch.nn.functional.l1_loss(heads.rootnet_forward(sd, x, kv, training=True) / 1000.0, gt), 1.0, g, 2e-4)


def test_full_two_iterations_golden(robot):
    """scripts/train_full.py:56-66 with full.yaml (clip 5), B = 2, two iterations (golden_full_2iter.npz)."""
    g = load("golden_full_2iter.npz")
    sd = full_sd()
    for k, v in sd.items():
        if v.dtype.is_floating_point and "running" not in k and not k.startswith("init_"):
            v.requires_grad_(True)
    rng = np.random.Generator(np.random.PCG64(2024))
    x_reg = torch.tensor(rng.integers(0, 256, (2, 3, 256, 256)).astype(np.float32)) / 255.
    x_root = torch.tensor(rng.integers(0, 256, (2, 3, 256, 256)).astype(np.float32)) / 255.
    K = torch.tensor(g["in:K"])
    bbox = g["in:bbox"]
    area = np.maximum(np.abs(bbox[:, 2] - bbox[:, 0]), np.abs(bbox[:, 3] - bbox[:, 1])) ** 2
    kv = torch.tensor(np.sqrt(g["in:K"][:, 0, 0] ** 2 * 1000.0 * 1000.0 / area).astype(np.float32))
    q, R, t = torch.tensor(g["in:q"]), torch.tensor(g["in:R"]), torch.tensor(g["in:t"])
    kp3d, kp2d, mask = torch.tensor(g["in:kp3d"]), torch.tensor(g["in:kp2d"]), torch.tensor(g["in:mask"])
    gt = dict(pose=q, root_rot=robot.get_rotation_at_specific_root(q, fk.rotmat_to_rot6d(R), t, root=3),
              root_trans=kp3d[:, 3], root_uv=kp2d[:, 3], kp3d=kp3d, kp2d=kp2d, mask=mask)
    _two_iterations(sd, lambda: heads.full_loss(heads.full_forward(sd, robot, x_reg, x_root, kv, K, training=True), gt, K)[0], 5.0, g, 2e-4)


SIM2REAL_CASES = {"iou_align": "mse_mean", "mse_mean": "mse_mean", "bce": "bce", "mse_sum": "mse_sum"}


@pytest.mark.parametrize("case", sorted(SIM2REAL_CASES))
def test_sim2real_mask_losses_golden(case):
    """The tensor-expression form of the render-and-compare losses (hrpe_amd.lib.core.function.sim2real_mask_loss on host
    tensors - what the fused kernel is tested against on the GPU) against the fixture written by executing the reference's own
    statements (scripts/train_sim2real.py:435-468; tests/golden/gen_golden.py sim2real_loss): terms, loss, autograd gradients."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import hrpe_amd  # noqa: F401
    from hrpe_amd.lib.core.function import sim2real_mask_loss
    g = load("golden_sim2real_loss.npz")
    r = torch.tensor(g["in:rendered"]).requires_grad_(True)
    a, b = torch.tensor(g["in:kp3d"]).requires_grad_(True), torch.tensor(g["in:kp3d_int"]).requires_grad_(True)
    wm, wi, ws, wa = [float(v) for v in g[f"{case}:weights"]]
    loss, terms = sim2real_mask_loss(r, torch.tensor(g["in:seg"]).unsqueeze(1), a, b, SIM2REAL_CASES[case],
                                     dict(mask=wm, iou=wi, scale=ws, align=wa))
    for k, v in terms.items():
        np.testing.assert_allclose(v.item(), g[f"{case}:{k}"], rtol=1e-6, err_msg=k)
    np.testing.assert_allclose(loss.item(), g[f"{case}:loss"], rtol=1e-6)
    loss.backward()
    np.testing.assert_allclose(r.grad.numpy(), g[f"{case}:d_rendered"], rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(a.grad.numpy(), g[f"{case}:d_kp3d"], rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(b.grad.numpy(), g[f"{case}:d_kp3d_int"], rtol=1e-5, atol=1e-9)


def test_mesh_pose_golden(robot):
    """oracle.fk.pose_mesh against the fixture of tests/golden/gen_golden.py mesh_pose: link poses from the reference's own
    kinematics, camera (R, T) recorded from the reference's get_rendered_mask_single_image_at_specific_root (roots 0 and 3, two
    samples behind the camera)."""
    g = load("golden_mesh_pose.npz")
    q, r6, t = torch.tensor(g["q"]), torch.tensor(g["rot6d"]), torch.tensor(g["t"])
    TL = fk.link_poses(robot.tree, q, fk.MESH_LINKS["panda"])
    np.testing.assert_allclose(TL[:, :, :3, :3].numpy(), g["link_R"], atol=2e-6)
    np.testing.assert_allclose(TL[:, :, :3, 3].numpy(), g["link_t"], atol=2e-6)
    for root in (0, 3):
        cam = fk.pose_mesh(robot, q, r6, t, torch.tensor(g["verts"]), torch.tensor(g["vert_link"]), root=root)
        np.testing.assert_allclose(cam.numpy(), g[f"cam_root{root}"], atol=5e-6, err_msg=f"root {root}")
    assert (g["T_root0"][[3, 7], 2] > 0).all() and (g["t"][[3, 7], 2] < 0).all()      # the mirrored samples are in the fixture


@pytest.mark.parametrize("tag", ["near", "far"])
def test_pose_loss_on_prescribed_predictions_golden(robot, tag):
    """The oracle's loss assembly (oracle/heads.py full_loss, restating lib/core/function.py:191-322) against the reference's own
    step function run on a stand-in model with prescribed outputs (tests/golden/gen_golden.py pose_loss): ten terms, the loss and
    autograd's gradient with respect to every prediction, for small AND large translation errors - `far` takes the branch of
    function.py:245-251 in which the translation term is damped by exp(-20 e)."""
    g = load("golden_pose_loss.npz")
    K = torch.tensor(g["in:K"])
    q, R, t = torch.tensor(g["in:q"]), torch.tensor(g["in:R"]), torch.tensor(g["in:t"])
    kp3d, kp2d, mask = torch.tensor(g["in:kp3d"]), torch.tensor(g["in:kp2d"]), torch.tensor(g["in:mask"])
    gt = dict(pose=q, root_rot=robot.get_rotation_at_specific_root(q, fk.rotmat_to_rot6d(R), t, root=3),
              root_trans=kp3d[:, 3], root_uv=kp2d[:, 3], kp3d=kp3d, kp2d=kp2d, mask=mask)
    pred = [torch.tensor(g[f"{tag}:pred:{n}"]).requires_grad_(True) for n in NAMES8]
    loss, terms = heads.full_loss(pred, gt, K)
    for k, v in terms.items():
        np.testing.assert_allclose(v.item(), g[f"{tag}:term:{k}"], rtol=2e-5, err_msg=k)
    np.testing.assert_allclose(loss.item(), g[f"{tag}:loss"], rtol=2e-5)
    loss.backward()
    for n, p in zip(NAMES8, pred):
        ref = g[f"{tag}:grad:{n}"]
        got = p.grad.numpy() if p.grad is not None else np.zeros_like(ref)
        np.testing.assert_allclose(got, ref, rtol=2e-4, atol=1e-7 + 1e-5 * np.abs(ref).max(), err_msg=n)


def _full_sd_quat():
    sd = {k: v for k, v in full_sd().items() if not k.startswith(("fc_rot_1", "decrot", "init_rot"))}
    sd.update(synth_state_dict({"fc_rot_1.weight": torch.empty(1024, 2048 + 4), "fc_rot_1.bias": torch.empty(1024),
                                "decrot.weight": torch.empty(4, 1024), "decrot.bias": torch.empty(4)}))
    sd["init_rot"] = torch.tensor([[1.0, 0.0, 0.0, 0.0]])          # rotmat_to_quat(identity), full_net.py:188-189
    return sd


def test_fk_quat_golden(robot):
    """Key-points, projection, gradients and the re-rooted rotation with the base-to-camera rotation as a quaternion
    (urdf_robot.py:86-92, 118-138; geometries.py:21-41, 63-82)."""
    g = load("golden_fk_quat.npz")
    q, r, t, K = [torch.tensor(g[k]) for k in ("q", "rot6d", "t", "K")]
    assert r.shape[1] == 4
    for root in (0, 3):
        tq, tr, tt = [x.clone().requires_grad_(True) for x in (q, r, t)]
        xyz = robot.get_keypoints_root(tq, tr, tt, root=root)
        uv = fk.project(K, xyz)
        np.testing.assert_allclose(xyz.detach().numpy(), g[f"xyz_root{root}"], atol=2e-6)
        np.testing.assert_allclose(uv.detach().numpy(), g[f"uv_root{root}"], atol=2e-3, rtol=1e-5)
        ((xyz * torch.tensor(g["w_xyz"])).sum() + (uv * torch.tensor(g["w_uv"])).sum()).backward()
        for name, x in (("gq", tq), ("grot", tr), ("gt", tt)):
            ref = g[f"{name}_root{root}"]
            np.testing.assert_allclose(x.grad.numpy(), ref, atol=2e-4 * max(1.0, np.abs(ref).max()), rtol=1e-3)
        rr = robot.get_rotation_at_specific_root(q, r, t, root=root)
        np.testing.assert_allclose(rr.numpy(), g[f"rootrot_root{root}"], atol=2e-6)


def test_full_eval_quat_golden(robot):
    """rotation_dim = 4 (full_net.py:129-131, 186-189): the 8-tuple with a quaternion as pred_rot."""
    g = load("golden_full_eval_quat.npz")
    x_reg, x_root, kv, K = synth_inputs(2)
    with torch.no_grad():
        out = heads.full_forward(_full_sd_quat(), robot, x_reg, x_root, kv, K)
    assert out[1].shape[1] == 4
    for n, t in zip(NAMES8, out):
        np.testing.assert_allclose(t.numpy(), g[n], atol=1e-5, rtol=1e-5, err_msg=n)


def test_segnet_oracle_resize_is_pillow_bit_for_bit():
    """oracle/segnet.py restates Pillow's 8-bit bicubic resize (the preprocessing of the mask network, reference
    lib/models/ctrnet/mask_inference.py:44-49).  Pillow is a dependency that IS installed here: the restatement is pinned against
    PIL.Image.resize itself, border rows / columns and odd sizes included.  (The network half of that oracle - torchvision's
    deeplabv3_resnet50 - has nothing to be pinned against in this image: parity unpinned, stated in its header.)"""
    from PIL import Image
    from oracle.segnet import pil_resize_half
    rng = np.random.default_rng(3)
    for H, W in ((480, 640), (64, 48), (22, 38), (9, 7)):
        a = rng.integers(0, 256, (H, W, 3)).astype(np.uint8)
        ref = np.asarray(Image.fromarray(a).resize((int(W * 0.5), int(H * 0.5))))
        assert np.array_equal(pil_resize_half(a), ref), (H, W)
    yy, xx = np.mgrid[0:120, 0:160]
    a = np.stack([yy * 2 % 256, xx % 256, (yy + xx) % 256], -1).astype(np.uint8)
    assert np.array_equal(pil_resize_half(a), np.asarray(Image.fromarray(a).resize((80, 60))))


def test_segnet_oracle_layer_plan_is_torchvisions_dilated_resnet50():
    """replace_stride_with_dilation = [False, True, True]: layer3 / layer4 keep stride 1, their first block keeps the previous
    dilation (torchvision resnet._make_layer)."""
    from oracle.segnet import layer_plan
    lp = dict(layer_plan())
    assert lp["layer1"] == [(1, 1)] * 3 and lp["layer2"] == [(2, 1)] + [(1, 1)] * 3
    assert lp["layer3"] == [(1, 1)] + [(1, 2)] * 5 and lp["layer4"] == [(1, 2)] + [(1, 4)] * 2
