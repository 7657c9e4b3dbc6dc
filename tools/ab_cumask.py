#!/usr/bin/env python3
"""A/B of lane PLACEMENT against lane time-sharing (run on the GPU box): the two trunk lanes of the training step on
HIP streams created with a CU mask (hipExtStreamCreateWithCUMask), each lane on its own half of the chip, against the same eager step
on ordinary streams.  Eager launches only: a captured graph's kernel nodes do not carry their capture stream's CU mask.
Prints ms per step for every placement."""
import ctypes as C
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from hrpe_amd.lib.core.function import compute_k_values, full_loss  # noqa: E402
from hrpe_amd.lib.utils.geometries import rotmat_to_rot6d  # noqa: E402
from hrpe_amd.optim import FusedClipAdam  # noqa: E402

DEV = torch.device("cuda:0")


def masked_stream(bits):
    """-> torch.cuda.ExternalStream on a HIP stream restricted to the CUs whose bit is set (256 bits, 8 words)."""
    import glob
    cand = glob.glob(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so*")) + ["libamdhip64.so"]
    hip = C.CDLL(cand[0])          # (the runtime torch already loaded: same handle)
    words = (C.c_uint32 * 8)(*[sum(((bits >> (32 * w + b)) & 1) << b for b in range(32)) for w in range(8)])
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), C.c_uint32(8), words)
    assert rc == 0, f"hipExtStreamCreateWithCUMask failed ({rc})"
    return torch.cuda.ExternalStream(st.value, device=DEV)


def main():
    B, steps = 64, int(os.environ.get("STEPS", "15"))
    torch.cuda.set_device(DEV)
    model = bench.build_model(0.5).to(DEV).set_compute_dtype(torch.bfloat16).train()
    opt = FusedClipAdam([p for p in model.parameters() if p.requires_grad], lr=1e-4, max_norm=5.0)
    d = {k: torch.tensor(v).to(DEV) for k, v in bench.synthetic_batch(B, 808).items()}
    K = d["K"]
    kv = compute_k_values(K[:, 0, 0], K[:, 1, 1], d["bbox"])
    rot6 = rotmat_to_rot6d(d["R"])
    with torch.no_grad():
        kp3d, kp2d = model.robot.get_keypoints_and_projection(d["q"], rot6, d["t"], K, root=0)
        gt = dict(pose=d["q"], root_rot=model.robot.get_rotation_at_specific_root(d["q"], rot6, d["t"], root=3),
                  root_trans=kp3d[:, 3].clone(), root_uv=kp2d[:, 3].clone(), kp3d=kp3d, kp2d=kp2d, mask=torch.ones(B, 7, device=DEV))
    from hrpe_amd import runtime
    runtime.GRAPH_CACHE = False          # eager launches throughout

    def step():
        loss, _ = full_loss(model(d["x_reg"], d["x_root"], kv, K), gt, K)
        loss.backward()
        opt.step()

    step()
    torch.cuda.synchronize()
    plan = next(iter(model._plans.values())).plan
    assert len(plan._side_streams) == 1, f"expected one side stream (a lane per trunk), got {len(plan._side_streams)}"
    plain_side = plan._side_streams[0]
    all_bits = (1 << 256) - 1
    xcd_lo = sum(1 << i for i in range(256) if i % 8 < 4)
    contig_lo = (1 << 128) - 1
    pack_plain = plan._pack_stream
    variants = [("ordinary streams", None, None),
                ("both lanes masked to ALL CUs (control)", all_bits, all_bits),
                ("lane 0: bits i % 8 < 4, lane 1: the rest (four XCDs each if the mask interleaves XCDs)", xcd_lo, all_bits ^ xcd_lo),
                ("lane 0: bits 0-127, lane 1: bits 128-255", contig_lo, all_bits ^ contig_lo),
                ("ordinary streams again", None, None)]
    for name, m0, m1 in variants:
        main_stream = masked_stream(m0) if m0 is not None else torch.cuda.Stream(DEV)
        plan._side_streams[0] = masked_stream(m1) if m1 is not None else plain_side
        with torch.cuda.stream(main_stream):
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / steps * 1e3
        print(f"{ms:8.2f} ms / step   {name}", flush=True)


if __name__ == "__main__":
    main()
