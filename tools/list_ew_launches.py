#!/usr/bin/env python3
"""development tool (GPU box): the element-wise launches of one EAGER training step of the benchmark network, grouped by
shape, each timed one by one with HIP events (one launch in flight at a time: the kernel's own duration, not its share of
the overlapped step).  Answers: which BatchNorm / activation passes still touch HBM, and how many bytes each moves."""
import collections
import os
import sys

os.environ.setdefault("HRP_SERIAL_LANES", "1")
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from hrpe_amd import _native as nv  # noqa: E402

EW = ("hrp_ew_fwd", "hrp_ew_bwd_reduce", "hrp_ew_bwd_apply")


def shape_of(fam, q):
    esz = 2 if q.dtype == nv.HRP_BF16 else 4
    if isinstance(q, nv.ConvDesc):      # a Bottleneck tail's backward pass carried out by a pointwise launch (tail_mode 3 / 4)
        px = q.N * q.Ho * q.Wo * esz
        return f"tail mode {q.tail_mode} C{q.Cin}>{q.Cout} @{q.Ho}", px * (q.Cin + q.Cout * (1 if q.tail_mode == 3 else 2 + bool(q.tail_side)))
    if fam == "hrp_ew_fwd":
        ins = "+".join(("bn" if q.inp[j].mode else "id") + (f"^{q.inp[j].up}" if q.inp[j].up != 1 else "") for j in range(q.nin))
        nbytes = q.N * q.H * q.W * q.C * esz * (1 + sum(1.0 / (q.inp[j].up ** 2) for j in range(q.nin)))
        return f"{ins} relu{q.relu}{' mask' if q.mask else ''} C{q.C} @{q.H}", nbytes
    up = q.inp.up
    npx = q.N * q.H * q.W * q.C * esz
    if fam == "hrp_ew_bwd_reduce":
        nbytes = npx + npx / up ** 2
    else:
        nbytes = npx + (2 + bool(q.din2)) * npx / up ** 2
    return (f"{'bn' if q.inp.mode else 'id'}{'^%d' % up if up != 1 else ''} relu{q.relu}{' mask' if q.mask else ''}"
            f"{' acc' if q.accumulate else ''}{' din2' if q.din2 else ''} C{q.C} @{q.H}"), nbytes


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
    torch.cuda.synchronize()
    rows = collections.defaultdict(lambda: [0, 0.0, 0.0])
    fam_tot = collections.defaultdict(lambda: [0, 0.0, 0.0])

    def hook(name, args, fn):
        fam, descs = bench.launch_descs(name, args)
        if fam not in EW:
            fn()
            return
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s = torch.cuda.current_stream()
        e0.record(s)
        fn()
        e1.record(s)
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3
        shapes, nb = [], 0.0
        for q in descs:
            sh, b = shape_of(fam, q)
            shapes.append(sh)
            nb += b
        key = (fam, len(descs), tuple(sorted(shapes)))
        r = rows[key]
        r[0] += 1
        r[1] += us
        r[2] += nb
        f = fam_tot[fam + (" (batched)" if len(descs) > 1 or name == "hrp_batch_launch" else "")]
        f[0] += 1
        f[1] += us
        f[2] += nb
    nv._profile_hook = hook
    out = model(d["x_reg"], d["x_root"], kv, K)
    loss, _ = full_loss(out, gt, K)
    loss.backward()
    nv._profile_hook = None
    for k, (n, us, nb) in sorted(fam_tot.items()):
        print(f"{k:32s} {n:4d} launches {us / 1e3:7.3f} ms {nb / 1e9:7.2f} GB  {nb / us / 1e6:5.2f} TB/s")
    for (fam, n, shapes), (c, us, nb) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
        print(f"{c:4d} x {fam[4:]:14s} [{n}] {us / c:7.1f} us {nb / c / 1e6:7.1f} MB {nb / us / 1e6:5.2f} TB/s  {', '.join(shapes)[:150]}")
