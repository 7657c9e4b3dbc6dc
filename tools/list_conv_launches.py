#!/usr/bin/env python3
"""development tool (GPU box): the conv-family launches of one training step of the benchmark network, grouped by shape."""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from hrpe_amd import _native as nv  # noqa: E402

if __name__ == "__main__":
    dev = torch.device("cuda:0")
    B = 64
    model = bench.build_model(0.5).to(dev).set_compute_dtype(torch.bfloat16).train()
    from hrpe_amd.lib.core.function import compute_k_values, full_loss
    from hrpe_amd.lib.utils.geometries import rotmat_to_rot6d
    d = {k: torch.tensor(v).to(dev) for k, v in bench.synthetic_batch(B, 4242).items()}
    K = d["K"]
    kv = compute_k_values(K[:, 0, 0], K[:, 1, 1], d["bbox"])
    rot6 = rotmat_to_rot6d(d["R"])
    with torch.no_grad():
        kp3d, kp2d = model.robot.get_keypoints_and_projection(d["q"], rot6, d["t"], K, root=0)
    gt = dict(pose=d["q"], root_rot=model.robot.get_rotation_at_specific_root(d["q"], rot6, d["t"], root=3), root_trans=kp3d[:, 3].clone(),
              root_uv=kp2d[:, 3].clone(), kp3d=kp3d, kp2d=kp2d, mask=torch.ones(B, 7, device=dev))
    for _ in range(2):
        loss, _ = full_loss(model(d["x_reg"], d["x_root"], kv, K), gt, K)
        loss.backward()
    phase = ["fwd"]
    rows = collections.Counter()

    def hook(name, args, fn):
        fam, descs = bench.launch_descs(name, args)
        if fam == "hrp_conv2d_fwd":
            key = (phase[0], len(descs), tuple(sorted(f"{q.Cin}>{q.Cout} t{q.ntaps} s{q.in_stride}/{q.out_stride} @{q.H}" for q in descs)))
            rows[key] += 1
        fn()
    nv._profile_hook = hook
    out = model(d["x_reg"], d["x_root"], kv, K)
    loss, _ = full_loss(out, gt, K)
    phase[0] = "bwd"
    loss.backward()
    nv._profile_hook = None
    tot = sum(rows.values())
    print("conv-family launches per step:", tot)
    for (ph, n, shapes), c in sorted(rows.items(), key=lambda kv: (-kv[1], kv[0])):
        print(f"{c:4d} x {ph} [{n} problem(s)] {', '.join(shapes)[:170]}")
