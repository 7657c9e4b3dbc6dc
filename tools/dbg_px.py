#!/usr/bin/env python3
"""Development aid (GPU box): where does the end-to-end pixel error of the bf16 eval pass come from?"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import bench  # noqa
from synth import synth_inputs, synth_state_dict
from hrpe_amd.lib.models.full_net import RootNetwithRegInt
from hrpe_amd.lib.utils.transforms import point_projection_from_3d_tensor
from hrpe_amd.lib.dataset.const import INITIAL_JOINT_ANGLE
dev = torch.device("cuda:0")
g = np.load(os.path.join(ROOT, "tests", "golden", "golden_full_eval.npz"))
init = {"robot_type": "panda", "pose_params": INITIAL_JOINT_ANGLE, "cam_params": np.eye(4), "init_pose_from_mean": True}
m = RootNetwithRegInt(init, bench.model_args(0.0))
m.load_state_dict(synth_state_dict(m.state_dict()))
m = m.to(dev).eval()
x_reg, x_root, kv, K = [t.to(dev) for t in synth_inputs(2)]
names = ["pose", "rot", "trans", "root_uv", "depth", "uvd", "xyz_int", "xyz_fk"]
ref = {n: torch.tensor(g[n]).to(dev) for n in names}
ref_uv = point_projection_from_3d_tensor(K, ref["xyz_fk"])
print("K", K[0].cpu().numpy().round(1).tolist(), "kv", kv.cpu().numpy())
print("ref xyz_fk[0]", ref["xyz_fk"][0].cpu().numpy().round(3).tolist())
for name, dt in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
    m.set_compute_dtype(dt)
    with torch.no_grad():
        o = m(x_reg, x_root, kv, K)
    print("==", name)
    for n, t in zip(names, o):
        e = (t - ref[n]).abs()
        print(f"  {n:8s} max abs err {e.max().item():.3e}  (scale {ref[n].abs().max().item():.3e})")
    uv = point_projection_from_3d_tensor(K, o[7])
    print("  px err per key-point:", (uv - ref_uv).abs().amax(-1).cpu().numpy().round(3).tolist())
    # which input of the FK drives it: swap single predictions for the reference's
    from hrpe_amd.lib.utils.urdf_robot import URDFRobot
    rb = m.robot
    for swap in ("pose", "rot", "trans"):
        a = {k: (ref[k] if k != swap else o[names.index(k)]) for k in ("pose", "rot", "trans")}
        xyz = rb.get_keypoints_root(a["pose"], a["rot"], a["trans"], root=3)
        uv2 = point_projection_from_3d_tensor(K, xyz)
        print(f"  only {swap} from the network: px err {(uv2 - ref_uv).abs().max().item():.4f}")
