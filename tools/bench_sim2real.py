#!/usr/bin/env python3
"""The self-supervised (render-and-compare) training step of BASELINE config 5 on synthetic data, timed on one GPU (development /
evidence tool - bench.py's contract is the supervised step).  Reference loop body: scripts/train_sim2real.py:139-146, 405-418, 435-468.

    python tools/bench_sim2real.py [--batch 32] [--steps 10] [--faces-per-side 14]

What is synthetic: the images, the robot mesh (one box per visual-mesh link, every side a grid of triangles: ~21 000 faces at the
default, the size of a real visual mesh set) and the weights of both networks.  The mask network (seg_mask_inference: device-side
PIL resize + DeepLabv3-ResNet50 + bilinear up-sampling + sigmoid, lib/models/ctrnet) RUNS in every step on [B, 3, 480, 640] images
as scripts/train_sim2real.py:412 does; with random weights its output is no silhouette, so the loss TARGET stays a mask rendered
once from a perturbed pose (--mask-target net uses the network's output instead: same work, meaningless loss).  The rasteriser
and the mask network are parity-UNPINNED (csrc/silhouette.hip; torchvision absent).  Plans replay from HIP graphs after their
second call (runtime.Runner)."""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import bench  # noqa: E402

DEV = torch.device("cuda:0")


def grid_box_mesh(n, seed=3):
    """One box per mesh link, every side an n x n grid of quads (2 n^2 triangles): (verts, vert_link, faces)."""
    g = np.random.Generator(np.random.PCG64(seed))
    verts, links, faces = [], [], []
    lin = np.linspace(-1.0, 1.0, n + 1, dtype=np.float32)
    for l in range(9):
        half = g.uniform(0.04, 0.09, 3).astype(np.float32)
        off = g.uniform(-0.03, 0.03, 3).astype(np.float32)
        for axis in range(3):
            for sign in (-1.0, 1.0):
                base = sum(len(v) for v in verts)
                a, b = np.meshgrid(lin, lin, indexing="ij")
                p = np.zeros(((n + 1) ** 2, 3), np.float32)
                p[:, axis] = sign
                p[:, (axis + 1) % 3] = a.reshape(-1)
                p[:, (axis + 2) % 3] = b.reshape(-1)
                verts.append(p * half + off)
                links += [l] * len(p)
                for i in range(n):
                    for j in range(n):
                        v00 = base + i * (n + 1) + j
                        faces += [(v00, v00 + 1, v00 + n + 2), (v00, v00 + n + 2, v00 + n + 1)]
    return (torch.tensor(np.concatenate(verts)), torch.tensor(np.asarray(links, np.uint8)), torch.tensor(np.asarray(faces, np.int32)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--faces-per-side", type=int, default=14)
    ap.add_argument("--mask-target", choices=["rendered", "net"], default="rendered")
    ap.add_argument("--no-mask-net", action="store_true")
    a = ap.parse_args()
    print(json.dumps(run(a)))


def run(a, check=False):
    """The step of main(), callable (tests/test_gpu_config5.py runs it once at the configuration's stated size).  check: also return
    what a test asserts on - the mask network's output shape, the rasteriser's largest per-pixel face count, gradient norms of both
    trunks and of the regressor heads."""
    from hrpe_amd.lib.core.function import compute_k_values, sim2real_mask_loss
    from hrpe_amd.optim import FusedClipAdam
    B = a.batch
    from synth import synth_state_dict
    model = bench.build_model(0.5)
    # (frozen BatchNorm normalises with the RUNNING statistics: the seeded state dict of the test fixtures carries plausible ones;
    # freshly initialised statistics of (0, 1) let the activations of a random network overflow)
    model.load_state_dict(synth_state_dict(model.state_dict()))
    model = model.to(DEV).set_compute_dtype(torch.bfloat16).train()
    for mod in model.modules():                     # scripts/train_sim2real.py:144-146
        if isinstance(mod, torch.nn.BatchNorm2d) or isinstance(mod, torch.nn.BatchNorm1d):
            mod.eval()
    d = {k: torch.tensor(v).to(DEV) for k, v in bench.synthetic_batch(B, 4242).items()}
    K = d["K"]
    kv = compute_k_values(K[:, 0, 0], K[:, 1, 1], d["bbox"])
    mesh = grid_box_mesh(a.faces_per_side)
    K_original = torch.tensor([[640.0, 0, 320.0], [0, 640.0, 240.0], [0, 0, 1.0]])
    renderer = model.robot.set_robot_renderer(K_original, original_image_size=(480, 640), scale=0.5, device=DEV, mesh=mesh)
    opt = FusedClipAdam([p for p in model.parameters() if p.requires_grad], lr=1e-6, max_norm=5.0)
    with torch.no_grad():
        pose, rot, trans = model(d["x_reg"], d["x_root"], kv, K)[:3]
        # random weights predict the robot anywhere: a constant per-sample offset puts key-point 3 at 1.5 m on the optical axis, so
        # that the silhouettes fill a realistic part of the image (the gradient still reaches the predicted translation)
        t_off = torch.tensor([0.0, 0.0, 1.5], device=DEV) - trans
        seg = model.robot.get_rendered_masks(pose, rot, trans + t_off + torch.tensor([0.02, -0.01, 0.03], device=DEV), renderer, root=3)
    weights = dict(mask=0.0, iou=1.0, scale=0.0, align=1.0)        # configs/panda/self_supervised/*.yaml:109-112
    seg_net = None
    if not a.no_mask_net:
        from hrpe_amd.lib.models.ctrnet.mask_inference import seg_mask_inference
        seg_net = seg_mask_inference((640.0, 640.0, 320.0, 240.0), "azure", allow_random_init=True)
        seg_net.load_state_dict(synth_state_dict(seg_net.state_dict()))
        seg_net = seg_net.to(DEV).set_compute_dtype(torch.bfloat16)
        images_original_255 = torch.randint(0, 256, (B, 3, 480, 640), device=DEV).float()      # train_sim2real.py:412
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(7)]
    acc = np.zeros(6)
    cover = 0.0

    def step(timed):
        nonlocal cover
        ev[0].record()
        target = seg
        if seg_net is not None:
            seg_masks = seg_net(images_original_255).detach()
            if a.mask_target == "net":
                target = seg_masks[:, 0]
        ev[1].record()
        out = model(d["x_reg"], d["x_root"], kv, K)
        ev[2].record()
        rendered = model.robot.get_rendered_masks(out[0], out[1], out[2] + t_off, renderer, root=3)
        ev[3].record()
        loss, _ = sim2real_mask_loss(rendered, target, out[7], out[6], "mse_mean", weights)
        ev[4].record()
        opt.zero_grad()
        loss.backward()
        ev[5].record()
        opt.step()
        ev[6].record()
        if timed:
            torch.cuda.synchronize()
            acc[:] += [ev[i].elapsed_time(ev[i + 1]) for i in range(6)]
            cover = float(rendered.detach().mean())
        return loss

    for _ in range(a.warmup):
        step(False)
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True)
    t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(a.steps):
        loss = step(True)
    t1.record()
    torch.cuda.synchronize()
    ms = t0.elapsed_time(t1) / a.steps
    names = ["mask network (resize + DeepLabv3-ResNet50 + upsample)", "network forward", "mesh posing + rasteriser", "mask losses",
             "backward (losses, rasteriser, network)", "clip + Adam"]
    extra = {}
    if check:
        with torch.no_grad():
            out = model(d["x_reg"], d["x_root"], kv, K)
            B_ = out[0].shape[0]
            model.robot.render_silhouette(out[0], out[1], out[2] + t_off, renderer.mesh, renderer.K.expand(B_, 3, 3), renderer.image_size, root=3,
                                          sigma=renderer.sigma, blur_radius=renderer.blur_radius, check_faces_per_pixel=True)
        gn = {}
        for name, pre in (("reg_trunk", "reg_backbone."), ("root_trunk", "rootnet_backbone."), ("heads", "fc_")):
            gs = [p.grad.float() for k, p in model.named_parameters() if k.startswith(pre) and p.grad is not None]
            gn[name] = float(torch.sqrt(sum((g_ * g_).sum() for g_ in gs))) if gs else 0.0
            gn[name + "_finite"] = all(bool(torch.isfinite(g_).all()) for g_ in gs)
        extra = {"grad_norms": gn, "max_faces_per_pixel": int(model.robot.last_faces_per_pixel),
                 "seg_mask_shape": list(seg_net(images_original_255).shape) if seg_net is not None else None,
                 "rendered_shape": list(seg.shape)}
    return dict({"workload": "self-supervised render-and-compare step (BASELINE config 5) on synthetic meshes / images, plans replayed "
                                  "from HIP graphs, full network bf16 with frozen BatchNorm, mask network on 480x640 images, 240x320 masks",
                      "batch": B, "images_per_sec": round(B / ms * 1e3, 1),
                      "ms_per_step": round(ms, 2), "phases_ms": {n: round(v / a.steps, 3) for n, v in zip(names, acc)},
                      "mesh": {"vertices": int(mesh[0].shape[0]), "faces": int(mesh[2].shape[0])}, "mask_coverage": round(cover, 4),
                      "loss": round(float(loss.detach()), 5), "rasteriser": "parity unpinned (csrc/silhouette.hip)",
                      "mask_network": ("absent (--no-mask-net)" if seg_net is None else
                                       "present (parity unpinned); loss target: " + ("its output" if a.mask_target == "net" else
                                                                                       "a mask rendered from a perturbed pose"))}, **extra)


if __name__ == "__main__":
    main()
