"""Generate the committed golden fixtures by running the REFERENCE itself on CPU.

Run by hand in the build container (needs /root/reference):  ``python tests/golden/gen_golden.py``
Writes small ``.npz`` files next to this script.  Weights come from ``synth.synth_state_dict`` and
images from seeded numpy streams, so only small inputs/outputs are stored.

Fixtures (reference call site that produced each):
  golden_fk.npz          URDFRobot.get_keypoints / get_keypoints_root / get_rotation_at_specific_root
                         (lib/utils/urdf_robot.py:82-199), point_projection_from_3d_tensor
                         (lib/utils/transforms.py:17-21), autograd grads; q=0 limb lengths.
  golden_fk_kuka.npz / golden_fk_baxter.npz   the same for the 7-DoF iiwa7 chain and the 15-DoF Baxter tree
                         (keypoint offsets of urdf_robot.py:57-74), generated with `fk_kuka fk_baxter`.
  golden_integral.npz    HeatmapIntegralPose.forward, hrnet branch (lib/utils/integral.py:97-186).
  golden_hrnet_eval.npz  PoseHighResolutionNet.forward eval, hm+feat (HRnet.py:499-570) + stage taps.
  golden_depthnet.npz    RootNet('hrnet32') eval forward, and train-mode L1 loss + grads
                         (lib/models/depth_net.py:92-137, scripts/train_depthnet.py:231-250).
  golden_full_eval.npz   RootNetwithRegInt.forward eval 8-tuple (lib/models/full_net.py:239-397).
  golden_full_eval_init.npz   the same with per-call init_pose / init_rot (full_net.py:245-248).
  golden_full_eval_fp64.npz   the same call with the reference's arithmetic in float64 (`full_eval_fp64`) + the pixel error of
                         the reference's own fp32 run against it: the noise floor the HIP fp32 path is held to.
  golden_full_eval_baxter.npz   the same for robot_type = "baxter" (15 DoF, 17 key-points), `full_eval_baxter`.
  golden_full_eval_resnet.npz / golden_full_train_resnet.npz   the same with backbone_name = "resnet50" (ResNet-50
                         trunk + deconv head of the shipped full.yaml), generated with `full_eval_resnet full_train_resnet`.
  golden_metrics.npz     lib/utils/metrics.py compute_metrics_batch (:8-113) and summary_add_pck (:116-162), `metrics`.
  golden_full_train.npz  lib/core/function.py farward_loss(train=True): loss terms + grads + BN
                         running stats after one step.
  golden_full_train_b8.npz   the same step at B = 8 (`full_train_b8`): tighter gradient tolerance (less BN noise).
  golden_depthnet_2iter.npz / golden_full_2iter.npz   TWO iterations of the trainers' loop - forward, loss, backward,
                         clip_grad_norm_, Adam(lr 1e-4).step - (scripts/train_depthnet.py:105, 231-250, 316-318 with
                         configs/panda/depthnet.yaml: B = 4, clip 1.0 = BASELINE.json configs[0]; scripts/train_full.py:42, 56-66
                         with full.yaml: clip 5.0, B = 2), `depthnet_2iter full_2iter`: losses of both iterations, the clipped
                         gradient norms, sampled parameter UPDATES and BatchNorm running statistics after the second step.
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as rh  # noqa: E402

rh.setup()
import torch  # noqa: E402
from synth import synth_inputs, synth_state_dict  # noqa: E402

torch.manual_seed(0)
torch.set_num_threads(8)

from lib.dataset.const import INITIAL_JOINT_ANGLE, JOINT_BOUNDS, JOINT_NAMES  # noqa: E402
from lib.models.backbones.HRnet import get_hrnet  # noqa: E402
from lib.models.depth_net import get_rootnet  # noqa: E402
from lib.models.full_net import RootNetwithRegInt  # noqa: E402
from lib.utils.geometries import rotmat_to_rot6d  # noqa: E402
from lib.utils.integral import HeatmapIntegralPose  # noqa: E402
from lib.utils.transforms import point_projection_from_3d_tensor  # noqa: E402
from lib.utils.urdf_robot import URDFRobot  # noqa: E402


def sample_indices(n, count, seed):
    g = np.random.Generator(np.random.PCG64(seed))
    return np.sort(g.choice(n, size=min(count, n), replace=False))


def summary(t, seed):
    """mean, abs-mean and 256 sampled elements of a tensor (fixture stays small)."""
    f = t.detach().reshape(-1).double()
    idx = sample_indices(f.numel(), 256, seed)
    return np.array([f.mean().item(), f.abs().mean().item()]), idx, f[idx].float().numpy()


def random_rotations(g, n):
    q = g.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = q.T
    R = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                  2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                  2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], 1)
    return R.reshape(n, 3, 3).astype(np.float32)


def gen_fk(robot_type="panda", other_root=3, quat=False):
    """quat: the base-to-camera rotation as an (unnormalised) quaternion (w, x, y, z) instead of the 6-D form
    (urdf_robot.py:86-92, 118-138; the rotation_dim == 4 variant of the network, full_net.py:186-189)."""
    robot = URDFRobot(robot_type)
    g = np.random.Generator(np.random.PCG64(1234))
    n = 256
    b = np.array(JOINT_BOUNDS[robot_type], dtype=np.float64)
    dof, nkp = len(b), len(robot.link_names)
    q = (b[:, 0] + (b[:, 1] - b[:, 0]) * g.random((n, dof))).astype(np.float32)
    rot6d = (random_rotations(g, n)[:, :2, :].reshape(n, 6)
             * g.uniform(0.5, 2.0, (n, 1)) + g.normal(0, 0.05, (n, 6))).astype(np.float32)
    if quat:
        from lib.utils.geometries import rotmat_to_quat
        rot6d = (rotmat_to_quat(torch.tensor(random_rotations(g, n).astype(np.float32))).numpy()
                 * g.uniform(0.5, 2.0, (n, 1)) + g.normal(0, 0.02, (n, 4))).astype(np.float32)
    t = np.stack([g.uniform(-.3, .3, n), g.uniform(-.3, .3, n), g.uniform(.6, 2.0, n)], 1).astype(np.float32)
    s = g.uniform(0.8, 2.5, n)
    K = np.zeros((n, 3, 3), np.float32)
    K[:, 0, 0] = 320 * s
    K[:, 1, 1] = 300 * s
    K[:, 0, 2] = 128 + g.uniform(-20, 20, n)
    K[:, 1, 2] = 128 + g.uniform(-20, 20, n)
    K[:, 2, 2] = 1
    out = dict(q=q, rot6d=rot6d, t=t, K=K)
    wx = g.normal(size=(n, nkp, 3)).astype(np.float32)
    wu = g.normal(size=(n, nkp, 2)).astype(np.float32) * 1e-2
    out["w_xyz"], out["w_uv"] = wx, wu
    out["roots"] = np.array([0, other_root])
    for root in (0, other_root):
        tq, tr, tt = [torch.tensor(a, requires_grad=True) for a in (q, rot6d, t)]
        xyz = robot.get_keypoints_root(tq, tr, tt, root=root)
        uv = point_projection_from_3d_tensor(torch.tensor(K), xyz)
        L = (xyz * torch.tensor(wx)).sum() + (uv * torch.tensor(wu)).sum()
        L.backward()
        out[f"xyz_root{root}"] = xyz.detach().numpy()
        out[f"uv_root{root}"] = uv.detach().numpy()
        out[f"gq_root{root}"] = tq.grad.numpy()
        out[f"grot_root{root}"] = tr.grad.numpy()
        out[f"gt_root{root}"] = tt.grad.numpy()
        with torch.no_grad():
            out[f"rootrot_root{root}"] = robot.get_rotation_at_specific_root(
                torch.tensor(q), torch.tensor(rot6d), torch.tensor(t), root=root).numpy()
    with torch.no_grad():
        out["fk_only"] = robot.get_keypoints_only_fk(torch.tensor(q)).numpy()
        out["fk_q0"] = robot.get_keypoints_only_fk(torch.zeros(1, dof)).numpy()
    name = "golden_fk.npz" if robot_type == "panda" else f"golden_fk_{robot_type}.npz"
    if quat:
        name = "golden_fk_quat.npz"
    np.savez_compressed(os.path.join(HERE, name), **out)
    print("fk ok", robot_type, out[f"xyz_root{other_root}"][0, :2])


def gen_fk_quat():
    gen_fk("panda", 3, quat=True)


def gen_fk_kuka():
    gen_fk("kuka", 3)


def gen_fk_baxter():
    gen_fk("baxter", 5)


def gen_metrics():
    """compute_metrics_batch (both call forms of function.py:139-168) and summary_add_pck (function.py:378) of the
    reference's lib/utils/metrics.py on seeded predictions; seaborn / matplotlib only draw curves there."""
    for name in ("seaborn", "matplotlib", "matplotlib.pyplot"):
        if name not in sys.modules:
            try:
                __import__(name)
            except ImportError:
                sys.modules[name] = types.ModuleType(name)
    from lib.utils.metrics import compute_metrics_batch, summary_add_pck
    robot = URDFRobot("panda")
    g = np.random.Generator(np.random.PCG64(77))
    B, nb = 16, 3
    b = np.array(JOINT_BOUNDS["panda"], dtype=np.float64)
    out, alldis, alldis_int = {}, {"dis3d": [], "dis2d": []}, {"dis3d": [], "dis2d": []}
    names = ["error3d", "error2d", "dis3d", "dis2d", "l1_jointerror", "mean_jointerror", "error_depth",
             "batch_error_relative", "error3d_relative"]
    for i in range(nb):
        q = (b[:, 0] + (b[:, 1] - b[:, 0]) * g.random((B, 8))).astype(np.float32)
        R = random_rotations(g, B)
        t = np.stack([g.uniform(-.4, .4, B), g.uniform(-.3, .3, B), g.uniform(.7, 2.0, B)], 1).astype(np.float32)
        K = np.tile(np.array([[615.0, 0, 320], [0, 615.0, 240], [0, 0, 1]], np.float32), (B, 1, 1))
        K[:, 0, 0] *= g.uniform(0.9, 1.1, B).astype(np.float32)
        rot6d = rotmat_to_rot6d(torch.tensor(R))
        with torch.no_grad():
            gt3d = robot.get_keypoints(torch.tensor(q), rot6d, torch.tensor(t))
            gt2d = point_projection_from_3d_tensor(torch.tensor(K), gt3d)
        scale = [0.002, 0.02, 0.2][i]                       # three error regimes so the AUC curves are not flat
        pq = q + g.normal(0, scale, q.shape).astype(np.float32)
        prot = rot6d.numpy() + g.normal(0, scale, (B, 6)).astype(np.float32)
        pt = t + g.normal(0, scale * 0.5, t.shape).astype(np.float32)
        pint = gt3d.numpy() + g.normal(0, scale * 0.3, gt3d.shape).astype(np.float32)
        with torch.no_grad():
            root_rot = robot.get_rotation_at_specific_root(torch.tensor(pq), torch.tensor(prot), torch.tensor(pt), root=3)
            root_t = robot.get_keypoints(torch.tensor(pq), torch.tensor(prot), torch.tensor(pt))[:, 3]
        ins = dict(gt3d=gt3d.numpy(), gt2d=gt2d.numpy(), K=K, q=q, pq=pq, prot=root_rot.numpy(), pt=root_t.numpy(), pint=pint)
        for k, v in ins.items():
            out[f"in{i}:{k}"] = v
        common = dict(robot=robot, gt_keypoints3d=gt3d, gt_keypoints2d=gt2d, K_original=torch.tensor(K),
                      gt_joint=torch.tensor(q), pred_depth=None, pred_xy=None, reference_keypoint_id=3)
        r = compute_metrics_batch(pred_joint=torch.tensor(pq), pred_rot=root_rot, pred_trans=root_t,
                                  pred_xyz_integral=None, **common)
        ri = compute_metrics_batch(pred_joint=None, pred_rot=None, pred_trans=None,
                                   pred_xyz_integral=torch.tensor(pint), **common)
        for n, a, c in zip(names, r, ri):
            out[f"fk{i}:{n}"] = np.asarray(a, dtype=np.float64)
            out[f"int{i}:{n}"] = np.asarray(c, dtype=np.float64)
        alldis["dis3d"].extend(list(r[0]))          # function.py:355-360: per-image means accumulate over the epoch
        alldis["dis2d"].extend(list(r[1]))
        alldis_int["dis3d"].extend(list(ri[0]))
        alldis_int["dis2d"].extend(list(ri[1]))
    for tag, ad in (("fk", alldis), ("int", alldis_int)):
        for k, v in summary_add_pck(ad).items():
            out[f"summary_{tag}:{k}"] = np.float64(v)
    np.savez_compressed(os.path.join(HERE, "golden_metrics.npz"), **out)
    print("metrics ok", {k: float(v) for k, v in out.items() if k.startswith("summary_fk:")})


def gen_integral():
    g = np.random.Generator(np.random.PCG64(4321))
    B = 2
    # logits: smooth bumps + noise so that the expectation is informative
    out = torch.as_tensor(g.normal(0, 2.0, (B, 7 * 64, 64, 64)).astype(np.float32))
    out[:, ::5] += 3.0
    _, _, kv, K = synth_inputs(B, seed=99)
    z = torch.tensor([[0.9], [1.4]])
    layer = HeatmapIntegralPose(backbone="hrnet32", num_joints=7, depth_dim=64, height_dim=64,
                                width_dim=64, norm_type="softmax", image_size=256.0,
                                bbox_3d_shape=[1300, 1300, 1300], rootid=3, fixroot=True)
    root_trans = torch.zeros(B, 3)
    root_trans[:, 2:3] = z
    xin = out.clone().requires_grad_(True)
    uvd, xyz = layer(xin, root_trans=root_trans, K=K)
    w = torch.as_tensor(g.normal(size=(B, 7, 3)).astype(np.float32))
    (uvd * w).sum().backward()
    gsum, gidx, gval = summary(xin.grad, 5)
    np.savez_compressed(os.path.join(HERE, "golden_integral.npz"), seed=4321, K=K.numpy(),
                        z_root=z.numpy(), uvd=uvd.detach().numpy(), xyz=xyz.detach().numpy(),
                        w=w.numpy(), g_summary=gsum, g_idx=gidx, g_val=gval)
    print("integral ok", uvd[0, 0])


def gen_hrnet_eval():
    m = get_hrnet(type_name=32, num_joints=7, depth_dim=64, pretrain=False, generate_feat=True,
                  generate_hm=True)
    m.load_state_dict(synth_state_dict(m.state_dict()))
    m.eval()
    x, _, _, _ = synth_inputs(2)
    taps = {}
    hooks = []
    for name in ["layer1", "stage2", "stage3", "stage4", "final_feat_layer"]:
        hooks.append(getattr(m, name).register_forward_hook(
            lambda mod, i, o, name=name: taps.__setitem__(name, o)))
    hooks.append(m.bn2.register_forward_hook(lambda mod, i, o: taps.__setitem__("stem_bn2", o)))
    with torch.no_grad():
        heat, feat = m(x)
    out = dict(feat=feat.numpy())
    s, idx, val = summary(heat, 11)
    out.update(heat_summary=s, heat_idx=idx, heat_val=val)
    flat = {"stem": torch.relu(taps["stem_bn2"]), "layer1": taps["layer1"],
            "head_map": taps["final_feat_layer"]}
    for st in ("stage2", "stage3", "stage4"):
        for b, t in enumerate(taps[st]):
            flat[f"{st}_{b}"] = t
    for k, t in flat.items():
        s, idx, val = summary(t, 17)
        out[f"tap_{k}_summary"], out[f"tap_{k}_idx"], out[f"tap_{k}_val"] = s, idx, val
    np.savez_compressed(os.path.join(HERE, "golden_hrnet_eval.npz"), **out)
    print("hrnet eval ok", feat[0, :4])


PICK_GRADS_DEPTHNET = [
    "backbone.conv1.weight", "backbone.bn1.weight", "backbone.layer1.0.conv2.weight",
    "backbone.layer1.0.downsample.0.weight", "backbone.transition1.1.0.0.weight",
    "backbone.stage2.0.branches.0.0.conv1.weight", "backbone.stage2.0.fuse_layers.0.1.0.weight",
    "backbone.stage2.0.fuse_layers.1.0.0.0.weight", "backbone.stage3.1.branches.1.2.conv2.weight",
    "backbone.stage3.2.branches.2.3.bn2.bias", "backbone.stage4.0.fuse_layers.3.0.1.0.weight",
    "backbone.stage4.2.branches.3.1.conv1.weight", "backbone.stage4.2.fuse_layers.0.3.1.weight",
    "backbone.incre_modules.2.0.conv3.weight", "backbone.downsamp_modules.1.0.weight",
    "backbone.downsamp_modules.1.0.bias", "backbone.final_feat_layer.0.weight",
    "backbone.final_feat_layer.1.weight", "depth_layer.weight", "depth_layer.bias",
]


def grad_fixture(model, names, out, tag):
    params = dict(model.named_parameters())
    for i, n in enumerate(names):
        g = params[n].grad
        s, idx, val = summary(g, 100 + i)
        out[f"{tag}grad:{n}:summary"], out[f"{tag}grad:{n}:idx"], out[f"{tag}grad:{n}:val"] = s, idx, val


def gen_depthnet():
    m = get_rootnet("hrnet32")
    m.load_state_dict(synth_state_dict(m.state_dict()))
    x, _, kv, _ = synth_inputs(2)
    out = {}
    m.eval()
    with torch.no_grad():
        out["depth_eval"] = m(x, kv).numpy()
    # one train_depthnet step: pred/1000 vs gt, L1 (train_depthnet.py:231-250)
    m.train()
    gt = torch.tensor([[1.1], [0.7]])
    pred = m(x, kv) / 1000.0
    loss = torch.nn.L1Loss()(pred, gt)
    loss.backward()
    out["depth_train"] = pred.detach().numpy()
    out["loss"] = np.array(loss.item())
    out["gt_depth"] = gt.numpy()
    grad_fixture(m, PICK_GRADS_DEPTHNET, out, "")
    sd = m.state_dict()
    for n in ["backbone.bn1.running_mean", "backbone.bn1.running_var",
              "backbone.stage4.2.branches.3.3.bn2.running_var",
              "backbone.final_feat_layer.1.running_mean"]:
        out["buf:" + n] = sd[n][:64].numpy()
    np.savez_compressed(os.path.join(HERE, "golden_depthnet.npz"), **out)
    print("depthnet ok", out["depth_eval"].ravel(), out["loss"])


def gen_depthnet_variants():
    """RootNet('hrnet32', use_offset=True, add_fc=True) (depth_net.py:44-70, 113-131): eval and one training step."""
    m = get_rootnet("hrnet32", use_offset=True, add_fc=True)
    m.load_state_dict(synth_state_dict(m.state_dict()))
    # B = 8: with two samples a BatchNorm1d output is sign(a - b) wherever |a - b| >> sqrt(eps) and anything in between
    # elsewhere - a 1e-5 relative perturbation of the pooled feature moved the B = 2 prediction by 4 %
    x, _, kv, _ = synth_inputs(8)
    out = {}
    m.eval()
    with torch.no_grad():
        out["depth_eval"] = m(x, kv).numpy()
    m.train()
    gt = torch.linspace(0.7, 1.4, 8).reshape(8, 1)
    pred = m(x, kv) / 1000.0
    loss = torch.nn.L1Loss()(pred, gt)
    loss.backward()
    out["depth_train"], out["loss"], out["gt_depth"] = pred.detach().numpy(), np.array(loss.item()), gt.numpy()
    grad_fixture(m, ["depth_fc1.weight", "depth_fc3.bias", "depth_bn2.weight", "depth_fc5.weight", "offset_layer.weight",
                     "depth_layer.weight", "backbone.final_feat_layer.0.weight", "backbone.conv1.weight"], out, "")
    sd = m.state_dict()
    for n in ["depth_bn1.running_mean", "depth_bn4.running_var"]:
        out["buf:" + n] = sd[n][:64].numpy()
    np.savez_compressed(os.path.join(HERE, "golden_depthnet_variants.npz"), **out)
    print("depthnet variants ok", out["depth_eval"].ravel(), out["depth_train"].ravel(), out["loss"])


def gen_depthnet_pred_xy():
    """RootNet('resnet50', pred_xy=True) (depth_net.py:33-43, 98-110, 133-135): [x, y, depth], eval + train forward
    at B = 4 and the gradients of a sum-of-squares loss on the x / y columns.  (The reference moves its index ramps to
    the GPU inside forward: .cuda() is a no-op for this CPU run.)"""
    from lib.models.backbones import Resnet as ref_resnet
    ref_resnet.ResNet.init_weights = lambda self, name: None
    torch.Tensor.cuda = lambda self, *a, **k: self
    m = get_rootnet("resnet50", pred_xy=True)
    m.load_state_dict(synth_state_dict(m.state_dict()))
    x, _, kv, _ = synth_inputs(4)
    out = {}
    m.eval()
    with torch.no_grad():
        out["coord_eval"] = m(x, kv).numpy()
    m.train()
    pred = m(x, kv)
    loss = (pred[:, :2] / 64.0).square().sum() + (pred[:, 2:] / 1000.0).square().sum()
    loss.backward()
    out["coord_train"], out["loss"] = pred.detach().numpy(), np.array(loss.item())
    grad_fixture(m, ["xy_layer.weight", "deconv_layers.0.weight", "deconv_layers.4.weight", "deconv_layers.6.weight",
                     "backbone.layer4.2.conv3.weight", "depth_layer.weight"], out, "")
    np.savez_compressed(os.path.join(HERE, "golden_depthnet_pred_xy.npz"), **out)
    print("depthnet pred_xy ok", out["coord_eval"], out["loss"])


def gen_depthnet_resnet():
    """DepthNet with a ResNet-50 trunk (depth_net.py:16-18, 93-95) and the full network with ResNet-50 for BOTH
    trunks, eval; one DepthNet training step (train_depthnet.py:231-250)."""
    from lib.models.backbones import Resnet as ref_resnet
    ref_resnet.ResNet.init_weights = lambda self, name: None
    m = get_rootnet("resnet50")
    m.load_state_dict(synth_state_dict(m.state_dict()))
    x, _, kv, _ = synth_inputs(2)
    out = {}
    m.eval()
    with torch.no_grad():
        out["depth_eval"] = m(x, kv).numpy()
    m.train()
    gt = torch.tensor([[1.1], [0.7]])
    pred = m(x, kv) / 1000.0
    loss = torch.nn.L1Loss()(pred, gt)
    loss.backward()
    out["depth_train"], out["loss"], out["gt_depth"] = pred.detach().numpy(), np.array(loss.item()), gt.numpy()
    grad_fixture(m, ["backbone.conv1.weight", "backbone.layer1.0.conv1.weight", "backbone.layer3.2.conv2.weight",
                     "backbone.layer4.0.downsample.0.weight", "backbone.layer4.2.bn3.weight", "depth_layer.weight"], out, "")
    args = rh.default_args()
    args.backbone_name = args.rootnet_backbone_name = "resnet50"
    init = {"robot_type": "panda", "pose_params": INITIAL_JOINT_ANGLE,
            "cam_params": np.eye(4, dtype=float), "init_pose_from_mean": True}
    full = RootNetwithRegInt(init, args)
    full.load_state_dict(synth_state_dict(full.state_dict()))
    full.eval()
    x_reg, x_root, kv, K = synth_inputs(2)
    with torch.no_grad():
        o = full(x_reg, x_root, kv, K)
    for n, t in zip(NAMES8, o):
        out["full:" + n] = t.numpy()
    np.savez_compressed(os.path.join(HERE, "golden_depthnet_resnet.npz"), **out)
    print("depthnet resnet ok", out["depth_eval"].ravel(), out["loss"])


def build_full(backbone_name=None, robot_type="panda", **over):
    args = rh.default_args(**over)
    if backbone_name is not None:
        # the shipped full.yaml pairs a ResNet regression trunk (+ deconv head) with the HRNet root trunk;
        # get_resnet() copies torchvision's ImageNet weights, which do not exist here and are overwritten anyway
        from lib.models.backbones import Resnet as ref_resnet
        ref_resnet.ResNet.init_weights = lambda self, name: None
        args.backbone_name = backbone_name
    init = {"robot_type": robot_type, "pose_params": INITIAL_JOINT_ANGLE,
            "cam_params": np.eye(4, dtype=float), "init_pose_from_mean": True}
    full = RootNetwithRegInt(init, args)
    full.load_state_dict(synth_state_dict(full.state_dict()))
    return full, args


NAMES8 = ["pose", "rot", "trans", "root_uv", "depth", "uvd", "xyz_int", "xyz_fk"]


def gen_full_eval():
    full, _ = build_full()
    full.eval()
    x_reg, x_root, kv, K = synth_inputs(2)
    with torch.no_grad():
        o = full(x_reg, x_root, kv, K)
    np.savez_compressed(os.path.join(HERE, "golden_full_eval.npz"),
                        **{n: t.numpy() for n, t in zip(NAMES8, o)})
    print("full eval ok", o[0][0, :3])


def gen_full_eval_fp64():
    """The reference's eval forward in float64 on the inputs of golden_full_eval.npz: the yardstick for what the reference's
    OWN fp32 arithmetic loses (forward casts with `.to(torch.float)` / `.float()`, lib/models/full_net.py:242-243,
    lib/utils/transforms.py:150-154: both are redirected to float64 for this one call).  Stores the fp64 8-tuple and the
    key-point pixel error of the reference's fp32 run against it (both projected with K, transforms.py:17-21)."""
    full, _ = build_full()
    full.eval()
    x_reg, x_root, kv, K = synth_inputs(2)
    with torch.no_grad():
        o32 = full(x_reg, x_root, kv, K)
    saved = (torch.float, torch.Tensor.float, torch.get_default_dtype(), torch.float32)
    try:
        torch.float = torch.float32 = torch.float64      # (integral.py:156 spells out dtype=torch.float32)
        torch.Tensor.float = lambda self, *a, **k: self.double()
        torch.set_default_dtype(torch.float64)
        full.double()
        for name, val in list(vars(full.robot).items()):      # plain tensor attributes of URDFRobot (key-point offsets)
            if torch.is_tensor(val) and val.is_floating_point():
                setattr(full.robot, name, val.double())
        with torch.no_grad():
            o64 = full(x_reg.double(), x_root.double(), kv.double(), K.double())
    finally:
        torch.float, torch.Tensor.float, torch.float32 = saved[0], saved[1], saved[3]
        torch.set_default_dtype(saved[2])
    assert all(t.dtype == torch.float64 for t in o64), [t.dtype for t in o64]
    uv32 = point_projection_from_3d_tensor(K.double(), o32[7].double())
    uv64 = point_projection_from_3d_tensor(K.double(), o64[7])
    px = (uv32 - uv64).abs().amax(-1)
    out = {n: t.numpy() for n, t in zip(NAMES8, o64)}
    out["ref_fp32_px_err"] = px.numpy()                                   # [B, key-points]
    out["ref_fp32_rel_err"] = np.array([float((a.double() - b).abs().max() / b.abs().max()) for a, b in zip(o32, o64)])
    np.savez_compressed(os.path.join(HERE, "golden_full_eval_fp64.npz"), **out)
    print("full eval fp64 ok: reference fp32 vs fp64 px err per key-point", px.numpy().round(5).tolist())
    print("   rel err of the 8-tuple:", dict(zip(NAMES8, out["ref_fp32_rel_err"].round(9).tolist())))


def gen_full_eval_init():
    """forward(..., init_pose=, init_rot=) (full_net.py:239, 245-248): the iterative regressors start from a per-sample pose /
    rotation the caller brings instead of the module's buffers."""
    full, _ = build_full()
    full.eval()
    x_reg, x_root, kv, K = synth_inputs(2)
    g = torch.Generator().manual_seed(77)
    init_pose = (torch.rand(2, full.init_pose.shape[1], generator=g) - 0.5) * 2.0
    a = torch.randn(2, 3, 3, generator=g)
    q, _ = torch.linalg.qr(a)
    init_rot = torch.cat([q[:, :, 0], q[:, :, 1]], dim=1)           # rot6d: the first two columns
    with torch.no_grad():
        o = full(x_reg, x_root, kv, K, init_pose=init_pose, init_rot=init_rot)
    np.savez_compressed(os.path.join(HERE, "golden_full_eval_init.npz"), init_pose=init_pose.numpy(), init_rot=init_rot.numpy(),
                        **{n: t.numpy() for n, t in zip(NAMES8, o)})
    print("full eval (per-call init) ok", o[0][0, :3])


def gen_full_eval_direct_rot():
    """direct_reg_rot = True (full_net.py:105-127, 333-345): the rotation comes from six stacked Linear layers."""
    full, _ = build_full(direct_reg_rot=True)
    full.eval()
    x_reg, x_root, kv, K = synth_inputs(2)
    with torch.no_grad():
        o = full(x_reg, x_root, kv, K)
    np.savez_compressed(os.path.join(HERE, "golden_full_eval_direct_rot.npz"),
                        **{n: t.numpy() for n, t in zip(NAMES8, o)})
    print("full eval (direct_reg_rot) ok", o[1][0])


def gen_full_eval_quat():
    """rotation_dim = 4 (full_net.py:129-131, 186-189): the rotation regressor iterates on a quaternion."""
    full, _ = build_full(rotation_dim=4)
    full.eval()
    x_reg, x_root, kv, K = synth_inputs(2)
    with torch.no_grad():
        o = full(x_reg, x_root, kv, K)
    assert o[1].shape[1] == 4
    np.savez_compressed(os.path.join(HERE, "golden_full_eval_quat.npz"), **{n: t.numpy() for n, t in zip(NAMES8, o)})
    print("full eval (quaternion) ok", o[1][0])


def gen_full_eval_multi_kp():
    """multi_kp = True, kps_need_depth = [0, 3, 6] (full_net.py:146-148, 275-279, 392-393): the 9-tuple."""
    full, _ = build_full(multi_kp=True, kps_need_depth=[0, 3, 6])
    full.eval()
    x_reg, x_root, kv, K = synth_inputs(2)
    with torch.no_grad():
        o = full(x_reg, x_root, kv, K)
    assert len(o) == 9
    names = NAMES8[:5] + ["depths"] + NAMES8[5:]
    np.savez_compressed(os.path.join(HERE, "golden_full_eval_multi_kp.npz"), **{n: t.numpy() for n, t in zip(names, o)})
    print("full eval (multi_kp) ok", o[4].ravel(), o[5])


def gen_full_eval_rot_matmul():
    """rot_iterative_matmul = True (full_net.py:346-362): the rotation estimate is updated by composing rotations."""
    full, _ = build_full(rot_iterative_matmul=True)
    full.eval()
    x_reg, x_root, kv, K = synth_inputs(2)
    with torch.no_grad():
        o = full(x_reg, x_root, kv, K)
    np.savez_compressed(os.path.join(HERE, "golden_full_eval_rot_matmul.npz"), **{n: t.numpy() for n, t in zip(NAMES8, o)})
    print("full eval (rot_iterative_matmul) ok", o[1][0])


def gen_full_add_fc():
    """add_fc = True (full_net.py:150-157, 261-270): eval 8-tuple at B = 2; one train-mode forward at B = 8 (the
    BatchNorm1d of the hour-glass MLP normalises over the batch: two samples would be a sign function)."""
    full, _ = build_full(add_fc=True)
    full.eval()
    x_reg, x_root, kv, K = synth_inputs(2)
    out = {}
    with torch.no_grad():
        o = full(x_reg, x_root, kv, K)
    for n, t in zip(NAMES8, o):
        out["eval:" + n] = t.numpy()
    full.train()
    x_reg, x_root, kv, K = synth_inputs(8)
    with torch.no_grad():
        o = full(x_reg, x_root, kv, K)
    out["train:depth"], out["train:trans"] = o[4].numpy(), o[2].numpy()
    out["buf:depth_bn.running_mean"] = full.state_dict()["depth_bn.running_mean"][:64].numpy()
    np.savez_compressed(os.path.join(HERE, "golden_full_add_fc.npz"), **out)
    print("full add_fc ok", out["eval:depth"].ravel(), out["train:depth"].ravel())


def gen_full_eval_joint_map():
    """reg_joint_map = True, joint_conv_dim = [128, 128, 128], ResNet-50 regression trunk (full_net.py:87-93, 218-237,
    313-316; HeatmapIntegralJoint, integral.py:186-232).  (The integral layer moves the joint bounds to the GPU inside
    forward: .cuda() is a no-op for this CPU run.)"""
    torch.Tensor.cuda = lambda self, *a, **k: self
    full, _ = build_full("resnet50", reg_joint_map=True, joint_conv_dim=[128, 128, 128])
    full.eval()
    x_reg, x_root, kv, K = synth_inputs(2)
    with torch.no_grad():
        o = full(x_reg, x_root, kv, K)
    np.savez_compressed(os.path.join(HERE, "golden_full_eval_joint_map.npz"), **{n: t.numpy() for n, t in zip(NAMES8, o)})
    print("full eval (reg_joint_map) ok", o[0][0])


def gen_full_eval_baxter():
    """15 DoF / 17 key-points: 1088 heat-map channels, 2063-wide pose regressor, tree FK with key-point offsets."""
    full, _ = build_full(robot_type="baxter")
    full.eval()
    x_reg, x_root, kv, K = synth_inputs(2)
    with torch.no_grad():
        o = full(x_reg, x_root, kv, K)
    np.savez_compressed(os.path.join(HERE, "golden_full_eval_baxter.npz"),
                        **{n: t.numpy() for n, t in zip(NAMES8, o)})
    print("full eval (baxter) ok", o[0][0, :3], o[7][0, :2])


def gen_full_eval_resnet():
    full, _ = build_full("resnet50")
    full.eval()
    x_reg, x_root, kv, K = synth_inputs(2)
    with torch.no_grad():
        o = full(x_reg, x_root, kv, K)
        x_out = full.reg_backbone(x_reg)
        heat = full.final_layer(full.deconv_layers(x_out))
    out = {n: t.numpy() for n, t in zip(NAMES8, o)}
    out["tap:x_out"] = x_out[:, ::64].numpy()         # every 64th channel of the trunk output
    out["tap:heat"] = heat[:, ::56, ::4, ::4].numpy()
    np.savez_compressed(os.path.join(HERE, "golden_full_eval_resnet.npz"), **out)
    print("full eval (resnet50 reg backbone) ok", o[0][0, :3])


def make_batch(B, robot, seed=2024):
    """A DreamDataset-shaped batch (lib/dataset/dream.py:393-413) from synthetic poses."""
    g = np.random.Generator(np.random.PCG64(seed))
    img_reg = g.integers(0, 256, (B, 3, 256, 256)).astype(np.float32)
    img_root = g.integers(0, 256, (B, 3, 256, 256)).astype(np.float32)
    b = np.array(JOINT_BOUNDS["panda"], dtype=np.float64)
    q = (b[:, 0] + (b[:, 1] - b[:, 0]) * g.random((B, 8))).astype(np.float32)
    R = random_rotations(g, B)
    t = np.stack([g.uniform(-.3, .3, B), g.uniform(-.3, .3, B), g.uniform(.8, 2.0, B)], 1).astype(np.float32)
    TCO = np.tile(np.eye(4, dtype=np.float32), (B, 1, 1))
    TCO[:, :3, :3] = R
    TCO[:, :3, 3] = t
    s = g.uniform(0.8, 2.5, B).astype(np.float32)
    K = np.zeros((B, 3, 3), np.float32)
    K[:, 0, 0] = K[:, 1, 1] = 320 * s
    K[:, 0, 2] = K[:, 1, 2] = 128
    K[:, 2, 2] = 1
    side = g.uniform(80, 240, B).astype(np.float32)
    bbox = np.stack([128 - side / 2, 128 - side / 2 * 0.8, 128 + side / 2, 128 + side / 2 * 0.8], 1).astype(np.float32)
    rot6d = rotmat_to_rot6d(torch.tensor(R))
    with torch.no_grad():
        kp3d = robot.get_keypoints(torch.tensor(q), rot6d, torch.tensor(t))
        kp2d = point_projection_from_3d_tensor(torch.tensor(K), kp3d)
    mask = np.ones((B, 7), np.float32)
    mask[0, 5] = 0.0
    jointpose = {n: [float(q[i, j]) for i in range(B)] for j, n in enumerate(JOINT_NAMES["panda"])}
    batch = {
        "root": {"images": torch.tensor(img_root), "K": torch.tensor(K),
                 "bbox_strict_bounded": torch.tensor(bbox), "bbox_gt2d_extended": torch.tensor(bbox)},
        "other": {"images": torch.tensor(img_reg), "K": torch.tensor(K), "keypoints_2d": kp2d,
                  "valid_mask_crop": torch.tensor(mask), "keypoints_3d": kp3d},
        "TCO": torch.tensor(TCO), "K_original": torch.tensor(K), "jointpose": jointpose,
        "keypoints_2d_original": kp2d.clone(), "valid_mask": torch.tensor(mask),
    }
    small = dict(q=q, R=R, t=t, K=K, bbox=bbox, kp3d=kp3d.numpy(), kp2d=kp2d.numpy(), mask=mask)
    return batch, small


PICK_GRADS_FULL = [
    "reg_backbone.conv1.weight", "reg_backbone.final_layer.weight", "reg_backbone.final_layer.bias",
    "reg_backbone.stage3.0.branches.0.1.conv1.weight", "reg_backbone.stage4.1.fuse_layers.2.0.0.0.weight",
    "reg_backbone.stage4.2.fuse_layers.0.1.0.weight", "reg_backbone.incre_modules.0.0.conv1.weight",
    "rootnet_backbone.conv1.weight", "rootnet_backbone.stage2.0.branches.1.3.bn2.weight",
    "rootnet_backbone.final_feat_layer.0.weight", "fc_pose_1.weight", "fc_pose_2.bias",
    "decpose.weight", "fc_rot_1.weight", "decrot.bias", "depth_layer.weight", "depth_layer.bias",
]


def import_reference_step_function():
    """lib/core/function.py pulls in cv2 / kornia / torchnet / tensorboard and BPnP.py:2 runs a CUDA
    op at import; none of that is on the synthetic-data path, so those modules are replaced by
    shells.  ``cast`` is the reference's own two-liner (lib/utils/utils.py:18-29: ``obj.to(device)``)."""
    bp = types.ModuleType("lib.utils.BPnP")
    bp.BPnP_m3d = None
    sys.modules["lib.utils.BPnP"] = bp
    mt = types.ModuleType("lib.utils.metrics")
    mt.compute_metrics_batch = mt.summary_add_pck = None
    sys.modules["lib.utils.metrics"] = mt
    ut = types.ModuleType("lib.utils.utils")
    ut.cast = lambda obj, device, dtype=None: obj.to(device)
    sys.modules["lib.utils.utils"] = ut
    tn = types.ModuleType("torchnet")
    tm = types.ModuleType("torchnet.meter")
    tm.AverageValueMeter = object
    sys.modules["torchnet"], sys.modules["torchnet.meter"] = tn, tm
    from lib.core import function
    return function


PICK_GRADS_RESNET = [
    "reg_backbone.conv1.weight", "reg_backbone.bn1.weight", "reg_backbone.layer1.0.conv1.weight",
    "reg_backbone.layer1.0.downsample.0.weight", "reg_backbone.layer2.0.conv2.weight",
    "reg_backbone.layer2.0.downsample.0.weight", "reg_backbone.layer3.5.conv3.weight", "reg_backbone.layer4.2.bn3.bias",
    "deconv_layers.0.weight", "deconv_layers.4.weight", "deconv_layers.6.weight", "final_layer.weight", "final_layer.bias",
    "rootnet_backbone.conv1.weight", "fc_pose_1.weight", "decrot.bias", "depth_layer.weight",
]


def gen_full_train(backbone_name=None, B=2, bn_eval=False, quat=False):
    function = import_reference_step_function()
    full, margs = build_full(backbone_name, **(dict(rotation_dim=4) if quat else {}))
    if bn_eval:
        # scripts/train_sim2real.py:139-146 (BASELINE config 5): the network trains with every BatchNorm module in eval() -
        # running statistics in the forward pass, gradients through them.  The step function calls model.train() itself, so
        # the switch is re-applied right after it.
        plain_train = full.train

        def train_with_frozen_bn(mode=True):
            plain_train(mode)
            for module in full.modules():
                if isinstance(module, (torch.nn.BatchNorm2d, torch.nn.BatchNorm1d)):
                    module.eval()
            return full
        full.train = train_with_frozen_bn
    batch, small = make_batch(B, full.robot)
    args = rh._AttrDict(dict(margs))
    args.update(urdf_robot_name="panda", use_origin_bbox=False, use_extended_bbox=True,
                train_ds_names="dream/synthetic/panda_synth_train_dr", use_joint_valid_mask=False,
                known_joint=False, joint_individual_weights=None, image_size=256.0, fix_mask=False,
                pose_loss_func="mse", rot_loss_func="mse", trans_loss_func="l2norm",
                depth_loss_func="l1", uv_loss_func="l2norm", kp2d_loss_func="l2norm",
                kp3d_loss_func="l2norm", kp2d_int_loss_func="l2norm", kp3d_int_loss_func="l2norm",
                align_3d_loss_func="l2norm", pose_loss_weight=1.0, rot_loss_weight=1.0,
                trans_loss_weight=1.0, depth_loss_weight=10.0, uv_loss_weight=1.0,
                kp2d_loss_weight=10.0, kp3d_loss_weight=10.0, kp2d_int_loss_weight=10.0,
                kp3d_int_loss_weight=10.0, align_3d_loss_weight=0.0)
    loss, terms = function.farward_loss(args, batch, full, full.robot, "cpu", [0], train=True)
    loss.backward()
    out = {"loss": np.array(loss.item())}
    for k, v in terms.items():
        out["term:" + k] = np.array(v.item())
    out.update({"in:" + k: v for k, v in small.items()})
    grad_fixture(full, PICK_GRADS_RESNET if backbone_name else PICK_GRADS_FULL, out, "")
    sd = full.state_dict()
    for n in (["reg_backbone.bn1.running_mean", "deconv_layers.7.running_var"] if backbone_name else
              ["reg_backbone.bn1.running_mean", "rootnet_backbone.stage4.2.branches.3.3.bn2.running_var"]):
        out["buf:" + n] = sd[n][:64].numpy()
    # the forward outputs of the same train-mode call, for localisation of a mismatch
    full.zero_grad()
    sd0 = synth_state_dict(full.state_dict())
    full.load_state_dict(sd0)
    full.train()
    x_reg = batch["other"]["images"].float() / 255.
    x_root = batch["root"]["images"].float() / 255.
    fx = small["K"][:, 0, 0]
    area = np.maximum(np.abs(small["bbox"][:, 2] - small["bbox"][:, 0]),
                      np.abs(small["bbox"][:, 3] - small["bbox"][:, 1])) ** 2
    kv = torch.tensor(np.sqrt(fx * fx * 1000.0 * 1000.0 / area).astype(np.float32))
    with torch.no_grad():
        o = full(x_reg, x_root, kv, torch.tensor(small["K"]))
    for n, t in zip(NAMES8, o):
        out["fwd:" + n] = t.numpy()
    out["k_values"] = kv.numpy()
    name = "golden_full_train_resnet.npz" if backbone_name else ("golden_full_train.npz" if B == 2 else f"golden_full_train_b{B}.npz")
    if bn_eval:
        name = "golden_full_train_bn_eval.npz"
    if quat:
        name = "golden_full_train_quat.npz"
    np.savez_compressed(os.path.join(HERE, name), **out)
    print("full train ok", name, out["loss"], {k: float(v) for k, v in terms.items()})


PICK_UPDATES_DEPTHNET = ["backbone.conv1.weight", "backbone.stage3.1.branches.0.2.conv2.weight", "backbone.stage4.0.fuse_layers.1.0.0.0.weight",
                         "backbone.stage2.0.branches.1.3.bn2.weight", "backbone.final_feat_layer.0.weight", "depth_layer.weight"]
PICK_UPDATES_FULL = ["reg_backbone.conv1.weight", "reg_backbone.stage3.0.branches.0.1.conv1.weight", "reg_backbone.final_layer.weight",
                     "rootnet_backbone.stage4.1.fuse_layers.2.0.0.0.weight", "fc_pose_1.weight", "decrot.bias", "depth_layer.weight"]


def _two_iterations(model, loss_fn, clip, picks, out):
    """The trainers' inner loop twice (scripts/train_full.py:56-66 / train_depthnet.py:316-322): zero_grad, loss, backward,
    clip_grad_norm_, Adam.step.  Records per iteration the loss and the total gradient norm clip_grad_norm_ returns, and after
    the second step the sampled parameter updates (p - p0) and running statistics."""
    params = dict(model.named_parameters())
    p0 = {n: params[n].detach().clone() for n in picks}
    idxs = {n: sample_indices(params[n].numel(), 256, 300 + i) for i, n in enumerate(picks)}
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, weight_decay=0.0)
    for it in range(2):
        opt.zero_grad()
        loss = loss_fn()
        loss.backward()
        norm = torch.nn.utils.clip_grad_norm_(model.parameters(), clip)
        # the (clipped) gradient Adam sees, at the sampled elements: the test leaves elements whose reference gradient is at
        # noise level (Adam's first steps are ~ lr * sign(g): their sign is not a property of the arithmetic) out of the count
        for n in picks:
            gr = params[n].grad.reshape(-1).double()
            out[f"grad{it + 1}:{n}:val"] = gr[idxs[n]].float().numpy()
            out[f"grad{it + 1}:{n}:absmean"] = np.array(gr.abs().mean().item())
        opt.step()
        out[f"loss{it + 1}"] = np.array(loss.item())
        out[f"grad_norm{it + 1}"] = np.array(float(norm))
    for i, n in enumerate(picks):
        upd = (params[n].detach() - p0[n]).reshape(-1).double()
        idx = idxs[n]
        out[f"upd:{n}:idx"], out[f"upd:{n}:val"] = idx, upd[idx].float().numpy()
        out[f"upd:{n}:absmean"] = np.array(upd.abs().mean().item())


def gen_depthnet_2iter():
    """BASELINE.json configs[0]: depthnet.yaml, B = 4, two iterations (random 256 x 256 images)."""
    m = get_rootnet("hrnet32")
    m.load_state_dict(synth_state_dict(m.state_dict()))
    x, _, kv, _ = synth_inputs(4)
    gt = torch.tensor([[1.1], [0.7], [1.6], [0.9]])
    m.train()
    out = {"gt_depth": gt.numpy()}
    _two_iterations(m, lambda: torch.nn.L1Loss()(m(x, kv) / 1000.0, gt), 1.0, PICK_UPDATES_DEPTHNET, out)
    sd = m.state_dict()
    for n in ["backbone.bn1.running_mean", "backbone.stage4.2.branches.3.3.bn2.running_var", "backbone.bn1.num_batches_tracked"]:
        out["buf:" + n] = sd[n].reshape(-1)[:64].numpy()
    np.savez_compressed(os.path.join(HERE, "golden_depthnet_2iter.npz"), **out)
    print("depthnet 2 iterations ok", {k: float(v) for k, v in out.items() if k.startswith(("loss", "grad_norm"))})


def gen_full_2iter():
    """scripts/train_full.py:56-66 with configs/panda/full.yaml (clip 5.0), B = 2, two iterations through the reference's own
    farward_loss."""
    function = import_reference_step_function()
    full, margs = build_full(None)
    batch, small = make_batch(2, full.robot)
    args = rh._AttrDict(dict(margs))
    args.update(urdf_robot_name="panda", use_origin_bbox=False, use_extended_bbox=True,
                train_ds_names="dream/synthetic/panda_synth_train_dr", use_joint_valid_mask=False,
                known_joint=False, joint_individual_weights=None, image_size=256.0, fix_mask=False,
                pose_loss_func="mse", rot_loss_func="mse", trans_loss_func="l2norm",
                depth_loss_func="l1", uv_loss_func="l2norm", kp2d_loss_func="l2norm",
                kp3d_loss_func="l2norm", kp2d_int_loss_func="l2norm", kp3d_int_loss_func="l2norm",
                align_3d_loss_func="l2norm", pose_loss_weight=1.0, rot_loss_weight=1.0,
                trans_loss_weight=1.0, depth_loss_weight=10.0, uv_loss_weight=1.0,
                kp2d_loss_weight=10.0, kp3d_loss_weight=10.0, kp2d_int_loss_weight=10.0,
                kp3d_int_loss_weight=10.0, align_3d_loss_weight=0.0)
    full.train()
    out = {"in:" + k: v for k, v in small.items()}
    _two_iterations(full, lambda: function.farward_loss(args, batch, full, full.robot, "cpu", [0], train=True)[0], 5.0,
                    PICK_UPDATES_FULL, out)
    sd = full.state_dict()
    for n in ["reg_backbone.bn1.running_mean", "rootnet_backbone.stage4.2.branches.3.3.bn2.running_var"]:
        out["buf:" + n] = sd[n][:64].numpy()
    np.savez_compressed(os.path.join(HERE, "golden_full_2iter.npz"), **out)
    print("full 2 iterations ok", {k: float(v) for k, v in out.items() if k.startswith(("loss", "grad_norm"))})


def gen_pose_loss():
    """The loss assembly of lib/core/function.py:191-322 on PRESCRIBED predictions: the reference's training-step function is run
    with a stand-in model whose forward returns seeded tensors (nn.Parameters, so that autograd leaves their gradients), once with
    small and once with large translation errors - both branches of the exp(-20 e) damping of function.py:245-251.  Pins
    hrp_pose_loss (terms + analytic gradient) directly against the reference instead of through the network."""
    function = import_reference_step_function()
    full, margs = build_full(None)
    B = 6
    batch, small = make_batch(B, full.robot)
    args = rh._AttrDict(dict(margs))
    args.update(urdf_robot_name="panda", use_origin_bbox=False, use_extended_bbox=True,
                train_ds_names="dream/synthetic/panda_synth_train_dr", use_joint_valid_mask=False,
                known_joint=False, joint_individual_weights=None, image_size=256.0, fix_mask=False,
                pose_loss_func="mse", rot_loss_func="mse", trans_loss_func="l2norm",
                depth_loss_func="l1", uv_loss_func="l2norm", kp2d_loss_func="l2norm",
                kp3d_loss_func="l2norm", kp2d_int_loss_func="l2norm", kp3d_int_loss_func="l2norm",
                align_3d_loss_func="l2norm", pose_loss_weight=1.0, rot_loss_weight=1.0,
                trans_loss_weight=1.0, depth_loss_weight=10.0, uv_loss_weight=1.0,
                kp2d_loss_weight=10.0, kp3d_loss_weight=10.0, kp2d_int_loss_weight=10.0,
                kp3d_int_loss_weight=10.0, align_3d_loss_weight=0.0)
    g = torch.Generator().manual_seed(99)
    kp3d = torch.tensor(small["kp3d"])
    out = {"in:" + k: v for k, v in small.items()}

    class Prescribed(torch.nn.Module):
        def __init__(self, tensors):
            super().__init__()
            self.p = torch.nn.ParameterList([torch.nn.Parameter(t.clone()) for t in tensors])

        def forward(self, reg_images, root_images, k_values, K=None):
            return tuple(p * 1.0 for p in self.p)

    r = lambda *sh: torch.randn(*sh, generator=g)      # noqa: E731
    for tag, spread in (("near", 0.05), ("far", 2.0)):
        preds = [torch.tensor(small["q"]) + 0.2 * r(B, 8), r(B, 6), kp3d[:, 3] + spread * r(B, 3),
                 torch.tensor(small["kp2d"])[:, 3] + 5.0 * r(B, 2), kp3d[:, 3, 2:3] + 0.05 * r(B, 1), r(B, 7, 3),
                 kp3d + 0.05 * r(B, 7, 3), kp3d + 0.05 * r(B, 7, 3)]
        model = Prescribed(preds)
        loss, terms = function.farward_loss(args, batch, model, full.robot, "cpu", [0], train=True)
        loss.backward()
        out[f"{tag}:loss"] = np.array(loss.item())
        for k, v in terms.items():
            out[f"{tag}:term:{k}"] = np.array(v.item())
        for n, p_, t0 in zip(NAMES8, model.p, preds):
            out[f"{tag}:pred:{n}"] = t0.numpy()
            out[f"{tag}:grad:{n}"] = (p_.grad if p_.grad is not None else torch.zeros_like(p_)).numpy()
        e = torch.norm(preds[2] - kp3d[:, 3], dim=1).mean().item()
        print("pose loss", tag, float(out[f"{tag}:loss"]), "mean translation error", e)
    np.savez_compressed(os.path.join(HERE, "golden_pose_loss.npz"), **out)


def gen_mesh_pose():
    """Mesh posing of the render-and-compare path (BASELINE config 5).  What the reference can run here: the link poses from its
    urdfpytorch kinematics (`robot.link_fk_batch`, the call inside URDFRobot.get_TWL) for the nine visual-mesh links of
    urdf_robot.py:209-219, and URDFRobot.get_rendered_mask_single_image_at_specific_root (urdf_robot.py:242-275) with the
    renderer replaced by a recorder of the camera (R, T) it is called with - pytorch3d, roboticstoolbox and the mesh files are
    not in the container.  The vertices are seeded points in the link frames, posed with the expression of
    mesh_renderer.py:148 (`verts @ R.T + t`) and viewed in pytorch3d's row-vector convention (`X @ R + T`)."""
    robot = URDFRobot("panda")
    g = np.random.Generator(np.random.PCG64(4321))
    n, V = 12, 96
    b = np.array(JOINT_BOUNDS["panda"], dtype=np.float64)
    q = (b[:, 0] + (b[:, 1] - b[:, 0]) * g.random((n, len(b)))).astype(np.float32)
    rot6d = (random_rotations(g, n)[:, :2, :].reshape(n, 6) * g.uniform(0.5, 2.0, (n, 1))).astype(np.float32)
    t = np.stack([g.uniform(-.3, .3, n), g.uniform(-.3, .3, n), g.uniform(.6, 2.0, n)], 1).astype(np.float32)
    t[3, 2], t[7, 2] = -0.9, -1.4                               # behind the camera: the mirrored branch of urdf_robot.py:250-251
    mesh_links = ["panda_link%d" % i for i in range(8)] + ["panda_hand"]
    verts = (g.normal(size=(V, 3)) * 0.08).astype(np.float32)
    vert_link = g.integers(0, len(mesh_links), V).astype(np.uint8)
    fk = robot.robot.link_fk_batch(torch.tensor(q), use_names=True)
    TL = torch.stack([fk[name] for name in mesh_links]).permute(1, 0, 2, 3)          # [n, 9, 4, 4]
    out = dict(q=q, rot6d=rot6d, t=t, verts=verts, vert_link=vert_link, link_R=TL[:, :, :3, :3].numpy(), link_t=TL[:, :, :3, 3].numpy())

    class Recorder:
        def silhouette_renderer(self, meshes_world=None, R=None, T=None):
            self.R, self.T = R.detach().clone(), T.detach().clone()
            return torch.zeros(1, 2, 2, 4)
    for root in (0, 3):
        cam = np.zeros((n, V, 3), np.float32)
        Rs, Ts = np.zeros((n, 3, 3), np.float32), np.zeros((n, 3), np.float32)
        for i in range(n):
            rec = Recorder()
            with torch.no_grad():
                robot.get_rendered_mask_single_image_at_specific_root(torch.tensor(q[i]), torch.tensor(rot6d[i]), torch.tensor(t[i]),
                                                                      None, rec, root=root)
            Rs[i], Ts[i] = rec.R[0].numpy(), rec.T[0].numpy()
            vl = torch.tensor(vert_link.astype(np.int64))
            posed = torch.einsum("vj,vkj->vk", torch.tensor(verts), TL[i, vl, :3, :3]) + TL[i, vl, :3, 3]     # verts_i @ R.T + t per link
            cam[i] = (posed @ rec.R[0] + rec.T[0]).numpy()
        out[f"cam_root{root}"], out[f"R_root{root}"], out[f"T_root{root}"] = cam, Rs, Ts
    np.savez_compressed(os.path.join(HERE, "golden_mesh_pose.npz"), **out)
    print("mesh pose ok", out["cam_root3"][0, :2], "flipped:", [int(i) for i in np.where(out["T_root0"][:, 2] > 0)[0] if t[i, 2] < 0])


def gen_sim2real_loss():
    """The mask / IoU / scale / 3-D alignment losses of the self-supervised trainer (BASELINE config 5).  They are inline code of
    scripts/train_sim2real.py (a 300-line closure that needs pytorch3d, a dataset and a segmentation checkpoint to run): the
    statements of lines 435-468 are read from the reference file HERE, at generation time, and executed on seeded inputs with the
    three loss modules of lines 409-411 - the reference's own arithmetic, with autograd's gradients."""
    import textwrap
    src = open(os.path.join(rh.REFERENCE_ROOT, "scripts", "train_sim2real.py")).read().split("\n")
    body = textwrap.dedent("\n".join(src[434:468]))          # 1-based lines 435..468
    assert body.lstrip().startswith("if args.mask_loss_func") and "loss = args.mask_loss_weight" in body
    g = torch.Generator().manual_seed(77)
    B, H, W, Kp = 6, 24, 32, 7
    out = {}
    cases = {"iou_align": ("mse_mean", (0.0, 1.0, 0.0, 1.0)), "mse_mean": ("mse_mean", (1.0, 0.5, 0.25, 2.0)),
             "bce": ("bce", (1.0, 1.0, 1.0, 1.0)), "mse_sum": ("mse_sum", (1.0, 0.0, 1.0, 0.0))}
    # silhouettes: soft blobs; two images are made to trip the scale filter (ratio > 5 and < 0.2)
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")

    def blob(cx, cy, rad, soft):
        return torch.sigmoid((rad - torch.sqrt((xx - cx) ** 2 + (yy - cy) ** 2)) / soft)
    seg = torch.stack([blob(16, 12, 7, 1.0), blob(10, 10, 5, 0.7), blob(20, 14, 9, 1.5), blob(16, 12, 10, 1.0), blob(16, 12, 3, 0.5),
                       blob(12, 8, 6, 1.0)])
    ren = torch.stack([blob(17, 12, 6.5, 0.8), blob(12, 11, 5, 0.7), blob(19, 13, 8, 1.2), blob(16, 12, 4, 0.8), blob(16, 12, 9, 1.0),
                       blob(13, 9, 6, 1.1)]).clamp(1e-4, 1 - 1e-4)
    kp_a = torch.randn(B, Kp, 3, generator=g) * 0.3
    kp_b = kp_a + torch.randn(B, Kp, 3, generator=g) * 0.02
    out["in:rendered"], out["in:seg"], out["in:kp3d"], out["in:kp3d_int"] = ren.numpy(), seg.numpy(), kp_a.numpy(), kp_b.numpy()
    for name, (func, (wm, wi, ws, wa)) in cases.items():
        rendered_masks = ren.clone().requires_grad_(True)
        pred_keypoints3d, pred_keypoints3d_int = kp_a.clone().requires_grad_(True), kp_b.clone().requires_grad_(True)
        ns = dict(torch=torch, args=rh._AttrDict(mask_loss_func=func, mask_loss_weight=wm, iou_loss_weight=wi, scale_loss_weight=ws,
                                                 align_3d_loss_weight=wa),
                  criterionBCE=torch.nn.BCELoss(), mse_sum=torch.nn.MSELoss(reduction="sum"), mse_mean=torch.nn.MSELoss(reduction="mean"),
                  rendered_masks=rendered_masks, seg_masks=seg.clone().unsqueeze(1), pred_keypoints3d=pred_keypoints3d,
                  pred_keypoints3d_int=pred_keypoints3d_int, cast=lambda obj, device, dtype=None: obj, device="cpu", loss_dict={})
        exec(body, ns)
        ns["loss"].backward()
        out[f"{name}:loss"] = np.array(ns["loss"].item())
        for k in ("loss_mask", "loss_iou", "loss_scale", "loss_error3d_align"):
            out[f"{name}:{k}"] = np.array(ns["loss_dict"][k].item())
        out[f"{name}:d_rendered"] = rendered_masks.grad.numpy()
        out[f"{name}:d_kp3d"] = pred_keypoints3d.grad.numpy()
        out[f"{name}:d_kp3d_int"] = pred_keypoints3d_int.grad.numpy()
        out[f"{name}:weights"] = np.array([wm, wi, ws, wa], dtype=np.float32)
        print("sim2real loss", name, float(out[f"{name}:loss"]), {k: float(out[f"{name}:{k}"]) for k in ("loss_mask", "loss_iou", "loss_scale", "loss_error3d_align")})
    np.savez_compressed(os.path.join(HERE, "golden_sim2real_loss.npz"), **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["fk", "integral", "hrnet_eval", "depthnet", "full_eval", "full_train"]
    for w in which:
        if w == "full_train_resnet":
            gen_full_train("resnet50")
        elif w == "full_train_quat":
            gen_full_train(None, B=2, quat=True)
        elif w == "full_train_bn_eval":
            gen_full_train(None, B=2, bn_eval=True)
        elif w == "full_train_b8":      # the same step at B = 8: train-mode BatchNorm over >= 512 samples per channel
            gen_full_train(None, B=8)   # amplifies rounding noise far less than B = 2, so gradients can be held tighter
        else:
            globals()["gen_" + w]()
