"""GPU: parity of the BENCHMARKED configuration (round 2).

* the reference's training step at B = 8 (fixture written by the reference itself) with gradients held to 1e-2;
* the bf16 trunk - the element type bench.py runs - against the reference's fp32 fixtures: eval 8-tuple, one training
  step (forward, loss terms, sampled gradients), with the measured errors asserted at ~3x the measurement (see the
  note at the tolerances for what bf16 can and cannot hold on this randomly weighted network);
* one fp32 training step at the benchmark's batch size B = 64 against the CPU oracle run on the host, and the bf16
  step against the fp32 HIP step on the same batch (conv tiling depends on N).
"""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import PANDA_URDF
from synth import synth_inputs
from test_gpu_model import DEV, NAMES8, build_full, load, summary_check

pytestmark = pytest.mark.gpu
GRAD_TOL_B8 = 1.5e-2         # fp32 gradients at B = 8: measured 7.5e-3 at most (round 2 gated 2.5e-2 on a noisy quantity)
# Measured on MI355X with the bf16 trunk against the reference's fp32 fixtures / the fp32 HIP path.  The step is
# bit-reproducible since round 3 (every gate sees ONE value, not a distribution), so the bounds are 2x the measurement
# (round 2: ~3x of the worst of several runs: uvd 0.6, root_uv 0.12, cosine 0.6).  Eval mode (running statistics) is tight.  Train mode normalises with the statistics of 8 (or 64)
# images of a randomly weighted 330-layer network: the soft-argmax over 262 144 nearly flat heat-map bins turns bf16
# rounding of the logits into visible uvd shifts, and gradients stored in bf16 lose the cancelling terms of the
# BatchNorm backward (g - mean(g) - xhat * mean(g * xhat)) - the heads' gradients stay within 10 %, the trunk's keep
# their direction (cosine), which is what the bounds below hold them to.
# measured eval:  pose 2.6e-4 rot 6.6e-4 trans 2.7e-4 root_uv 9.4e-4 depth 2.2e-4 uvd 4.9e-3 xyz_int 2.5e-3 xyz_fk 5.5e-4
BF16_EVAL_TOL = {"pose": 5e-4, "rot": 1.3e-3, "trans": 5e-4, "root_uv": 1.5e-3, "depth": 4e-4, "uvd": 1e-2, "xyz_int": 5e-3, "xyz_fk": 1.1e-3}
# measured B = 8: pose 1.9e-3 rot 4.0e-3 trans 1.0e-2 root_uv 3.3e-2 depth 2.0e-3 uvd 0.151 xyz_int 2.4e-2 xyz_fk 1.0e-2
BF16_TRAIN_TOL = {"pose": 4e-3, "rot": 8e-3, "trans": 2e-2, "root_uv": 6.5e-2, "depth": 4e-3, "uvd": 0.3, "xyz_int": 5e-2, "xyz_fk": 2e-2}
# measured B = 64: pose 3.0e-3 rot 5.6e-3 trans 4.5e-2 root_uv 9.5e-2 depth 8.7e-3 uvd 0.25 xyz_int 6.2e-2 xyz_fk 4.5e-2
BF16_B64_TOL = {"pose": 6e-3, "rot": 1.1e-2, "trans": 9e-2, "root_uv": 0.19, "depth": 1.8e-2, "uvd": 0.5, "xyz_int": 0.125, "xyz_fk": 9e-2}
HEADS = ("fc_pose", "fc_rot", "decpose", "decrot", "depth_layer")


# bf16 gradients against fp32, PER TENSOR (a global cosine gate of 0.65 said "not broken", not "parity").  The step is
# bit-reproducible, so every sampled tensor has ONE value; measured on MI355X in round 6 (tests run with -s print them):
#   (l2 error, cosine) at B = 8 against the reference's fixture / at B = 64 against the fp32 HIP step of the same batch.
# Gates: l2 <= 1.5 x the measurement, 1 - cosine <= 1.5 x the measurement (+ 0.002).  What the numbers say is unchanged: the heads'
# gradients are within 2-4 % (15 % at B = 64, where the loss is dominated by one term), the trunk keeps the direction (0.75-0.99) and
# loses length - gradients stored in bf16 lose the cancelling terms of the BatchNorm backward on this randomly weighted network.
BF16_GRAD_MEASURED = {
    "bf16 B=8": {
        "reg_backbone.conv1.weight": (0.628, 0.816), "reg_backbone.final_layer.weight": (0.192, 0.989),
        "reg_backbone.final_layer.bias": (0.0795, 0.998), "reg_backbone.stage3.0.branches.0.1.conv1.weight": (0.565, 0.836),
        "reg_backbone.stage4.1.fuse_layers.2.0.0.0.weight": (0.423, 0.907), "reg_backbone.stage4.2.fuse_layers.0.1.0.weight": (0.153, 0.988),
        "reg_backbone.incre_modules.0.0.conv1.weight": (0.482, 0.878), "rootnet_backbone.conv1.weight": (0.669, 0.774),
        "rootnet_backbone.stage2.0.branches.1.3.bn2.weight": (0.756, 0.746), "rootnet_backbone.final_feat_layer.0.weight": (0.267, 0.966),
        "fc_pose_1.weight": (0.0352, 1.0), "fc_pose_2.bias": (0.0269, 1.0), "decpose.weight": (0.034, 1.0), "fc_rot_1.weight": (0.0338, 1.0),
        "decrot.bias": (0.0257, 1.0), "depth_layer.weight": (0.0185, 1.0), "depth_layer.bias": (0.0141, 1.0)},
    "bf16 B=64": {
        "reg_backbone.conv1.weight": (0.684, 0.809), "reg_backbone.stage3.0.branches.0.1.conv1.weight": (0.601, 0.842),
        "reg_backbone.stage4.2.branches.3.3.conv2.weight": (0.488, 0.910), "reg_backbone.stage4.1.fuse_layers.2.0.0.0.weight": (0.534, 0.883),
        "reg_backbone.final_layer.weight": (0.152, 0.989), "rootnet_backbone.stage2.0.branches.1.3.bn2.weight": (0.780, 0.769),
        "rootnet_backbone.final_feat_layer.0.weight": (0.375, 0.951), "fc_pose_1.weight": (0.144, 1.0), "decrot.bias": (0.146, 1.0),
        "depth_layer.weight": (0.0606, 1.0)}}


def _check_bf16_grads(l2, cos, what):
    table = BF16_GRAD_MEASURED[what]
    assert set(l2) <= set(table), f"{what}: no measurement recorded for {sorted(set(l2) - set(table))}"
    for n, e in l2.items():
        m_l2, m_cos = table[n]
        assert e <= 1.5 * m_l2, f"{what} {n}: l2 err {e} (measured {m_l2})"
        assert 1.0 - cos[n] <= 1.5 * (1.0 - m_cos) + 2e-3, f"{what} {n}: cosine {cos[n]} (measured {m_cos})"


def _train_step_inputs(g, m, B):
    """Inputs / ground truth of the reference's training-step fixtures (tests/golden/gen_golden.py::make_batch)."""
    from hrpe_amd.lib.core.function import compute_k_values
    from hrpe_amd.lib.utils.geometries import rotmat_to_rot6d
    rng = np.random.Generator(np.random.PCG64(2024))
    x_reg = (torch.tensor(rng.integers(0, 256, (B, 3, 256, 256)).astype(np.float32)) / 255.).to(DEV)
    x_root = (torch.tensor(rng.integers(0, 256, (B, 3, 256, 256)).astype(np.float32)) / 255.).to(DEV)
    K = torch.tensor(g["in:K"]).to(DEV)
    kv = compute_k_values(K[:, 0, 0], K[:, 1, 1], torch.tensor(g["in:bbox"]).to(DEV))
    q, R, t = [torch.tensor(g[k]).to(DEV) for k in ("in:q", "in:R", "in:t")]
    kp3d, kp2d, mask = [torch.tensor(g[k]).to(DEV) for k in ("in:kp3d", "in:kp2d", "in:mask")]
    gt = dict(pose=q, root_rot=m.robot.get_rotation_at_specific_root(q, rotmat_to_rot6d(R), t, root=3),
              root_trans=kp3d[:, 3], root_uv=kp2d[:, 3], kp3d=kp3d, kp2d=kp2d, mask=mask)
    return x_reg, x_root, kv, K, gt


def _sampled_grad_errors(g, params):
    errs = {}
    for key in g.files:
        if key.startswith("grad:") and key.endswith(":val"):
            name = key.split(":")[1]
            val, idx = g[f"grad:{name}:val"], g[f"grad:{name}:idx"]
            if np.abs(val).max() < 1e-8:
                continue
            got = params[name].grad.reshape(-1)[torch.tensor(idx).to(DEV)].cpu().numpy()
            errs[name] = float(np.linalg.norm(got - val) / np.linalg.norm(val))
    return errs


def test_full_train_step_golden_b8():
    """The reference's training step at B = 8 (golden_full_train_b8.npz).  Train-mode BatchNorm over >= 512 samples per
    channel amplifies rounding less than the B = 2 fixture (gate 3e-2 there): measured over repeated runs on MI355X the
    l2 error of the sampled gradients is <= 1e-5 for the heads, 5e-3 .. 9e-3 for most trunk tensors and 1.0e-2 .. 1.6e-2
    for the two deepest ones (stem conv1, stage3.0 - the end of a 330-layer backward chain whose run-to-run spread from
    the fp32 atomics alone is 5e-3 .. 1e-2, see test_grouped_plan_matches_one_by_one_plan).  Gate: 2.5e-2 on the l2 error,
    with summary_check's allowance for the few elements a ReLU tie moves."""
    from hrpe_amd.lib.core.function import full_loss
    g = load("golden_full_train_b8.npz")
    m = build_full().train()
    x_reg, x_root, kv, K, gt = _train_step_inputs(g, m, 8)
    np.testing.assert_allclose(kv.cpu().numpy(), g["k_values"], rtol=1e-6)
    pred = m(x_reg, x_root, kv, K)
    for n, p in zip(NAMES8, pred):
        ref = g["fwd:" + n]
        err = np.abs(p.detach().cpu().numpy() - ref).max() / (np.abs(ref).max() + 1e-12)
        assert err < 5e-4, f"train fwd {n}: rel err {err}"
    loss, terms = full_loss(pred, gt, K)
    for k, v in terms.items():
        np.testing.assert_allclose(v.item(), g["term:" + k], rtol=1e-3, atol=1e-8, err_msg=k)
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=5e-4)
    loss.backward()
    params = dict(m.named_parameters())
    errs = _sampled_grad_errors(g, params)
    print("\nB=8 fp32 gradient l2 err:", {k: f"{v:.2e}" for k, v in errs.items()})
    for key in g.files:
        if key.startswith("grad:") and key.endswith(":val"):
            summary_check(params[key.split(":")[1]].grad, g, f"grad:{key.split(':')[1]}:", GRAD_TOL_B8, what="full B=8 ")


def _bf16_train_step(m):
    """One bf16 training step at B = 8 against golden_full_train_b8.npz; raises AssertionError on a violated bound."""
    from hrpe_amd.lib.core.function import full_loss
    g = load("golden_full_train_b8.npz")
    m.train()
    x_reg, x_root, kv, K, gt = _train_step_inputs(g, m, 8)
    pred = m(x_reg, x_root, kv, K)
    errs = {n: float(np.abs(p.detach().cpu().numpy() - g["fwd:" + n]).max() / (np.abs(g["fwd:" + n]).max() + 1e-12))
            for n, p in zip(NAMES8, pred)}
    print("\nbf16 train fwd rel err:", {k: f"{v:.2e}" for k, v in errs.items()})
    train_errs = errs
    loss, terms = full_loss(pred, gt, K)
    terr = {k: abs(v.item() - float(g["term:" + k])) / (abs(float(g["term:" + k])) + 1e-12) for k, v in terms.items()}
    print("\nbf16 loss-term rel err:", {k: f"{v:.2e}" for k, v in terr.items()}, "loss", loss.item(), float(g["loss"]))
    for k, e in terr.items():
        if float(g["term:" + k]) > 1e-3:      # (loss_trans is ~2e-6 on this batch: exp(-20 e) damped, function.py:245-251)
            assert e < 3e-2, f"bf16 loss term {k}: rel err {e}"
    assert abs(loss.item() - float(g["loss"])) < 1e-2 * float(g["loss"])
    loss.backward()
    gerr = _sampled_grad_errors(g, dict(m.named_parameters()))
    print("\nbf16 gradient l2 err:", {k: f"{v:.2e}" for k, v in gerr.items()})
    cos = {}
    params = dict(m.named_parameters())
    for key in g.files:
        if key.startswith("grad:") and key.endswith(":val"):
            name = key.split(":")[1]
            val, idx = g[f"grad:{name}:val"], g[f"grad:{name}:idx"]
            if np.abs(val).max() < 1e-8:
                continue
            got = params[name].grad.reshape(-1)[torch.tensor(idx).to(DEV)].cpu().numpy()
            cos[name] = float(np.dot(got, val) / (np.linalg.norm(got) * np.linalg.norm(val) + 1e-30))
    print("\nbf16 gradient cosine:", {k: f"{v:.3f}" for k, v in cos.items()})
    for n, e in train_errs.items():
        assert e < BF16_TRAIN_TOL[n], f"bf16 train fwd {n}: rel err {e}"
    _check_bf16_grads(gerr, cos, "bf16 B=8")


def test_full_bf16_against_reference_fixtures():
    """bf16 trunk vs the reference: eval 8-tuple (golden_full_eval.npz) and one training step at B = 8
    (golden_full_train_b8.npz: forward 8-tuple, loss terms, sampled gradients)."""
    from hrpe_amd.lib.core.function import full_loss
    g = load("golden_full_eval.npz")
    m = build_full().eval().set_compute_dtype(torch.bfloat16)
    x_reg, x_root, kv, K = synth_inputs(2)
    with torch.no_grad():
        out = m(x_reg.to(DEV), x_root.to(DEV), kv.to(DEV), K.to(DEV))
    errs = {n: float(np.abs(t.cpu().numpy() - g[n]).max() / (np.abs(g[n]).max() + 1e-12)) for n, t in zip(NAMES8, out)}
    print("\nbf16 eval rel err:", {k: f"{v:.2e}" for k, v in errs.items()})
    for n, e in errs.items():
        assert e < BF16_EVAL_TOL[n], f"bf16 eval {n}: rel err {e}"
    # (Round 2 re-measured a violated bound once on a fresh model: the step was noisy - fp32 statistic atomics -> bf16 rounding
    # -> ReLU / arg-max ties.  It is bit-reproducible now, the bounds hold or fail for good.)
    _bf16_train_step(m)


def test_full_train_step_b64_against_oracle_and_bf16():
    """One fp32 training step at B = 64 against the CPU oracle on the host, then bf16 against fp32 HIP on the same batch."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import fk as ofk, heads as oheads
    from hrpe_amd.lib.core.function import full_loss
    B = 64
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    m = build_full().train()
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    for k, v in sd.items():
        if v.dtype.is_floating_point and "running" not in k and not k.startswith("init_"):
            v.requires_grad_(True)
    rng = np.random.Generator(np.random.PCG64(64))
    x_reg = torch.tensor(rng.random((B, 3, 256, 256), dtype=np.float32))
    x_root = torch.tensor(rng.random((B, 3, 256, 256), dtype=np.float32))
    s = rng.uniform(0.8, 2.5, B).astype(np.float32)
    K = torch.zeros(B, 3, 3)
    K[:, 0, 0] = K[:, 1, 1] = torch.tensor(320.0 * s)
    K[:, 0, 2] = K[:, 1, 2] = 128.0
    K[:, 2, 2] = 1.0
    kv = torch.tensor(rng.uniform(1500.0, 6000.0, B).astype(np.float32))
    b = np.array([[-2.9, 2.9]] * 7 + [[0.0, 0.04]])
    q = torch.tensor((b[:, 0] + (b[:, 1] - b[:, 0]) * rng.random((B, 8))).astype(np.float32))
    Rm = torch.linalg.qr(torch.tensor(rng.normal(size=(B, 3, 3)).astype(np.float32)))[0]
    Rm = Rm * torch.sign(torch.linalg.det(Rm)).reshape(B, 1, 1)
    t = torch.tensor(np.stack([rng.uniform(-.3, .3, B), rng.uniform(-.3, .3, B), rng.uniform(.8, 2.0, B)], 1).astype(np.float32))
    robot = ofk.Robot(PANDA_URDF)
    rot6 = ofk.rotmat_to_rot6d(Rm)
    with torch.no_grad():
        kp3d = robot.get_keypoints(q, rot6, t)
        kp2d = ofk.project(K, kp3d)
        gt = dict(pose=q, root_rot=robot.get_rotation_at_specific_root(q, rot6, t, root=3), root_trans=kp3d[:, 3],
                  root_uv=kp2d[:, 3], kp3d=kp3d, kp2d=kp2d, mask=torch.ones(B, 7))
    pred_o = oheads.full_forward(sd, robot, x_reg, x_root, kv, K, training=True)
    loss_o, _ = oheads.full_loss(pred_o, gt, K)
    loss_o.backward()
    gt_d = {k: v.to(DEV) for k, v in gt.items()}
    dev = [v.to(DEV) for v in (x_reg, x_root, kv, K)]

    def step():
        m.zero_grad()
        pred = m(*dev)
        loss, _ = full_loss(pred, gt_d, dev[3])
        loss.backward()
        return [p.detach().clone() for p in pred], loss.item(), {n: p.grad.clone() for n, p in m.named_parameters()}
    names = ["reg_backbone.conv1.weight", "reg_backbone.stage3.0.branches.0.1.conv1.weight",
             "reg_backbone.stage4.2.branches.3.3.conv2.weight", "reg_backbone.stage4.1.fuse_layers.2.0.0.0.weight",
             "reg_backbone.final_layer.weight", "rootnet_backbone.stage2.0.branches.1.3.bn2.weight",
             "rootnet_backbone.final_feat_layer.0.weight", "fc_pose_1.weight", "decrot.bias", "depth_layer.weight"]
    rel = lambda a, r: float((a - r).norm() / (r.norm() + 1e-30))   # noqa: E731
    pred32, loss32, grads32 = step()
    for n, p, r in zip(NAMES8, pred32, pred_o):
        e = float((p.cpu() - r.detach()).abs().max() / (r.detach().abs().max() + 1e-12))
        assert e < 1e-3, f"B=64 fp32 fwd {n}: rel err {e}"
    assert abs(loss32 - loss_o.item()) < 1e-3 * abs(loss_o.item()), (loss32, loss_o.item())
    g32 = {n: rel(grads32[n].cpu(), sd[n].grad) for n in names}
    print("\nB=64 fp32 vs oracle: loss", loss32, loss_o.item(), "gradient l2 err", {k: f"{v:.2e}" for k, v in g32.items()})
    assert max(g32.values()) < 1e-2, g32
    m.set_compute_dtype(torch.bfloat16)
    pred16, loss16, grads16 = step()
    e16 = {n: float((p - r).abs().max() / (r.abs().max() + 1e-12)) for n, p, r in zip(NAMES8, pred16, pred32)}
    g16 = {n: rel(grads16[n], grads32[n]) for n in names}
    print("\nB=64 bf16 vs fp32 HIP: fwd rel err", {k: f"{v:.2e}" for k, v in e16.items()}, "loss", loss16, loss32,
          "gradient l2 err", {k: f"{v:.2e}" for k, v in g16.items()})
    c16 = {n: float((grads16[n] * grads32[n]).sum() / (grads16[n].norm() * grads32[n].norm() + 1e-30)) for n in names}
    print("\nB=64 bf16 gradient cosine:", {k: f"{v:.3f}" for k, v in c16.items()})
    for n, e in e16.items():
        assert e < BF16_B64_TOL[n], f"B=64 bf16 fwd {n}: rel err {e}"
    assert abs(loss16 - loss32) < 1e-2 * abs(loss32)
    _check_bf16_grads(g16, c16, "bf16 B=64")
