"""Helper of tests/test_gpu_round3.py::test_two_rank_whole_step (not a test module): ONE rank of a 2-rank data-parallel
training step on a shared GPU (gloo), the replacement of the reference's nn.DataParallel step (lib/core/function.py:100-102,
scripts/train_full.py:53-67).  Checks, on every rank:
  * the arena after the all-reduce == the mean of the two ranks' own gradients (gathered before the reduction), bit for bit;
  * the split backward (first part | all-reduce of the final ranges | rest) gives the same arena as the plain backward;
  * after FusedClipAdam.step() both ranks hold identical parameters and BatchNorm buffers stay per-replica."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from hrpe_amd.lib.core.function import compute_k_values, full_loss  # noqa: E402
from hrpe_amd.lib.utils.geometries import rotmat_to_rot6d  # noqa: E402
from hrpe_amd.optim import FusedClipAdam  # noqa: E402
from hrpe_amd.parallel import GradAllReducer, broadcast_module, init_distributed  # noqa: E402


def main():
    rank, world, _ = init_distributed(backend="gloo")
    assert world == 2
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    B = 4
    model = bench.build_model(0.0).to(dev).set_compute_dtype(torch.bfloat16).train()
    if rank == 1:                                   # a replica that starts elsewhere: the broadcast must align it with rank 0
        with torch.no_grad():
            for p in model.parameters():
                p.mul_(1.01)
    broadcast_module(model)
    params = [p for p in model.parameters() if p.requires_grad]
    opt = FusedClipAdam(params, lr=1e-4, max_norm=5.0)
    d = {k: torch.tensor(v).to(dev) for k, v in bench.synthetic_batch(B, 808 + rank).items()}     # per-rank data
    K = d["K"]
    kv = compute_k_values(K[:, 0, 0], K[:, 1, 1], d["bbox"])
    rot6 = rotmat_to_rot6d(d["R"])
    with torch.no_grad():
        kp3d, kp2d = model.robot.get_keypoints_and_projection(d["q"], rot6, d["t"], K, root=0)
        gt = dict(pose=d["q"], root_rot=model.robot.get_rotation_at_specific_root(d["q"], rot6, d["t"], root=3),
                  root_trans=kp3d[:, 3].clone(), root_uv=kp2d[:, 3].clone(), kp3d=kp3d, kp2d=kp2d,
                  mask=torch.ones(B, 7, device=dev))

    def fwd_bwd():
        loss, _ = full_loss(model(d["x_reg"], d["x_root"], kv, K), gt, K)
        loss.backward()
        return loss.detach()

    # ---- plain step: backward, all-reduce of the whole arena -----------------------------------------------------
    fwd_bwd()
    (arena,) = model.flat_grads()
    torch.cuda.synchronize(dev)
    own = arena.clone()
    both = [torch.empty_like(own) for _ in range(2)]
    dist.all_gather(both, own)
    GradAllReducer(bucket_mb=64)([arena])
    torch.cuda.synchronize(dev)
    mean = (both[0] + both[1]) / 2
    assert torch.equal(arena, mean), f"rank {rank}: all-reduced arena != mean of the ranks' gradients ({(arena - mean).abs().max().item()})"
    assert (both[0] - both[1]).abs().max().item() > 0, "the two ranks computed the same gradient: per-rank data missing"
    reduced_plain = arena.clone()
    # the SAME bf16 step again: bit for bit (fp64 statistic slots, ordered split reductions - round 2 measured a run-to-run
    # spread of 0.4 of the gradient norm here at B = 4 and compared "equal" against three times that)
    fwd_bwd()
    torch.cuda.synchronize(dev)
    assert torch.equal(arena, own), f"rank {rank}: the same step twice gives different gradients ({((arena - own).norm() / own.norm()).item()})"

    # ---- the same step with the backward split around the all-reduce of the final ranges (bench.py at N > 1) -----------
    sp = model.enable_split_backward()
    assert sp is not None, "no split found"
    plan, final = sp
    fwd_bwd()                                       # stops at the split
    model.check_split_backward(final)               # runs the rest; raises if it touched the final ranges
    # the split backward == the plain backward, bit for bit (the same launches in the same order on every lane)
    torch.cuda.synchronize(dev)
    assert torch.equal(arena, own), f"rank {rank}: split backward differs from the plain backward by {((arena - own).norm() / own.norm()).item()}"
    red = GradAllReducer(bucket_mb=64)
    fwd_bwd()                                       # first part again
    w = red.start(arena, final)                     # final ranges travel ...
    plan.run_backward("rest")                       # ... while the rest runs
    w += red.start(arena, GradAllReducer.complement(final, arena.numel()))
    red.finish(w, [arena])
    torch.cuda.synchronize(dev)
    assert torch.equal(arena, reduced_plain), \
        f"rank {rank}: overlapped all-reduce differs from the plain one by {((arena - reduced_plain).norm() / reduced_plain.norm()).item()}"
    other_a = [torch.empty_like(arena) for _ in range(2)]
    dist.all_gather(other_a, arena)
    assert torch.equal(other_a[0], other_a[1]), "the ranks hold different averaged gradients after the overlapped all-reduce"
    model.disable_split_backward()

    # ---- k-way split (bench.py at N > 1): the heads', stage 4's, stage 3's gradients travel while the next segment runs -----
    sp = model.enable_split_backward(fracs=(0.25, 0.5, 0.8, 0.9))
    assert sp is not None, "no cuts found"
    plan, groups = sp
    assert len(plan.bwd_cuts) >= 2 and len(groups) == len(plan.bwd_cuts), (plan.bwd_cuts, len(groups))      # (after stage 4, after stage 3)
    covered = [r for grp in groups for r in grp]
    tail = GradAllReducer.complement(covered, arena.numel())
    tail_frac = sum(n for _, n in tail) / arena.numel()
    assert tail_frac <= 0.20, f"the collective left behind the backward carries {tail_frac:.2f} of the bytes"
    fwd_bwd()                                       # segment 0
    model.check_split_backward(groups)              # runs segments 1 .. k; raises if one touched an earlier group's ranges
    torch.cuda.synchronize(dev)
    assert torch.equal(arena, own), f"rank {rank}: {len(groups) + 1}-segment backward differs from the plain backward"
    fwd_bwd()
    w = red.start(arena, groups[0])
    for j in range(1, len(plan.bwd_cuts) + 1):
        plan.run_backward(("seg", j))
        if j < len(groups):
            w += red.start(arena, groups[j])
    w += red.start(arena, tail)
    red.finish(w, [arena])
    torch.cuda.synchronize(dev)
    assert torch.equal(arena, reduced_plain), \
        f"rank {rank}: {len(groups)}-cut overlapped all-reduce differs from the plain one by {((arena - reduced_plain).norm() / reduced_plain.norm()).item()}"
    # bf16 payload (HRP_GRAD_PAYLOAD=bf16): half the bytes on the links, the mean within bf16 rounding of the fp32 one
    redh = GradAllReducer(bucket_mb=64, payload="bf16")
    arena.copy_(own)
    redh([arena])
    torch.cuda.synchronize(dev)
    rel = ((arena - reduced_plain).norm() / reduced_plain.norm()).item()
    assert 0 < rel < 8e-3, f"rank {rank}: bf16-payload all-reduce is {rel} away from the fp32 one"
    model.disable_split_backward()

    # ---- optimizer step on the averaged gradients: replicas stay identical ---------------------------------------------
    arena.copy_(reduced_plain)

    def same_on_both(t, what):
        both_t = [torch.empty_like(t) for _ in range(2)]
        dist.all_gather(both_t, t.contiguous())
        assert torch.equal(both_t[0], both_t[1]), f"rank {rank}: {what} differ between the ranks ({(both_t[0] - both_t[1]).abs().max().item()})"
    same_on_both(torch.cat([p.detach().reshape(-1) for p in params]), "parameters before the step")
    same_on_both(arena, "averaged gradients")
    opt.step()
    torch.cuda.synchronize(dev)
    same_on_both(opt._slots, "gradient-norm slots")
    same_on_both(torch.cat([p.detach().reshape(-1) for p in params]), "parameters after the step")
    rm = torch.cat([b.reshape(-1).float() for n, b in model.named_buffers() if n.endswith("running_mean")])
    rms = [torch.empty_like(rm) for _ in range(2)]
    dist.all_gather(rms, rm)
    assert not torch.equal(rms[0], rms[1]), "BatchNorm running statistics are per replica (no SyncBN, as under DataParallel)"
    print(f"rank {rank}: ok ({arena.numel()} gradient elements, split final fraction {sum(n for _, n in final) / arena.numel():.2f}, "
          f"{len(groups)} cuts at {plan.bwd_cuts} leave a tail of {tail_frac:.3f}, bf16 payload within {rel:.1e}, "
          f"the step repeats bit for bit)")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
