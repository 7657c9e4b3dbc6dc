#!/usr/bin/env python3
"""development tool (GPU box): the conv-family launches of one training step of the benchmark network, grouped by shape."""
import collections
import os
import sys

os.environ.setdefault("HRP_SERIAL_LANES", "1")
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
    EVAL = "--eval" in sys.argv
    if EVAL:
        model.eval()
    for _ in range(2):
        if EVAL:
            with torch.no_grad():
                model(d["x_reg"], d["x_root"], kv, K)
            continue
        loss, _ = full_loss(model(d["x_reg"], d["x_root"], kv, K), gt, K)
        loss.backward()
    phase = ["fwd"]
    rows = collections.Counter()

    times, nbytes = collections.Counter(), collections.Counter()

    def hook(name, args, fn):
        fam, descs = bench.launch_descs(name, args)
        if fam not in ("hrp_conv2d_fwd", "hrp_block_launch"):
            fn()
            return
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s = torch.cuda.current_stream()
        e0.record(s)
        fn()
        e1.record(s)
        torch.cuda.synchronize()
        if fam == "hrp_block_launch":      # fused inference BasicBlocks (hrp_block_desc: conv1, conv2)
            key = (phase[0], len(descs), tuple(sorted(f"block {q.conv1.Cin} @{q.conv1.H}" for q in descs)))
        else:
            key = (phase[0], len(descs), tuple(sorted(f"{q.Cin}>{q.Cout} t{q.ntaps} s{q.in_stride}/{q.out_stride} @{q.H}"
                                                     + (" pro%d" % q.pro_mode if q.pro_mode else "") + (" bnb" if q.bnb_x else "") + (" res" if q.res else "") for q in descs)))
        rows[key] += 1
        times[key] += e0.elapsed_time(e1) * 1e3
        nbytes[key] += bench.conv_bytes(name, args, extended=True)
    nv._profile_hook = hook
    if EVAL:
        with torch.no_grad():
            model(d["x_reg"], d["x_root"], kv, K)
    else:
        out = model(d["x_reg"], d["x_root"], kv, K)
        loss, _ = full_loss(out, gt, K)
        phase[0] = "bwd"
        loss.backward()
    nv._profile_hook = None
    tot = sum(rows.values())
    print("conv-family launches per step:", tot)
    print(f"one by one (HIP events, serial lanes): {sum(times.values()) / 1e3:.2f} ms, {sum(nbytes.values()) / 1e9:.1f} GB incl. fused operands")
    for key, c in sorted(rows.items(), key=lambda kv: -times[kv[0]]):
        ph, n, shapes = key
        print(f"{c:4d} x {ph} [{n}] {times[key] / c:7.1f} us {nbytes[key] / c / 1e6:7.1f} MB {nbytes[key] / times[key] / 1e6:5.2f} TB/s  {', '.join(shapes)[:150]}")
