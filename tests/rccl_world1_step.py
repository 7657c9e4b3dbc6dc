"""Helper of tests/test_gpu_rccl.py (not a test module): the data-parallel training step with a REAL RCCL process group of one
rank (backend "nccl", HRP_DIST_WORLD1=1) - the only execution of the collective path a one-GPU box can produce (VERDICT r5 item 6;
the reference's counterpart is nn.DataParallel's gather / scatter, lib/core/function.py:100-102).

Two models with identical weights train side by side for ITERS iterations:
  A  the unsplit step: one HIP graph = forward + loss + backward, then clip + Adam (what bench.py runs at N = 1)
  B  the k-cut step of bench.py at N > 1: graph (forward + backward segment 0) | all_reduce(async) of the ranges that became final,
     on RCCL's stream | graph (segment 1) | all_reduce | ... | all_reduce of the tail | wait, / world | graph (clip + Adam)
After every iteration B's reduced gradient arena must equal A's bit for bit (a reduction over one rank is the identity, so any
difference is an ordering hazard between the graph replays on the compute stream and the collectives on RCCL's stream), and at the
end the parameters must be identical."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["HRP_DIST_WORLD1"] = "1"
os.environ.setdefault("WORLD_SIZE", "1")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("LOCAL_RANK", "0")
import bench  # noqa: E402
from hrpe_amd.lib.core.function import compute_k_values, full_loss  # noqa: E402
from hrpe_amd.lib.utils.geometries import rotmat_to_rot6d  # noqa: E402
from hrpe_amd.optim import FusedClipAdam  # noqa: E402
from hrpe_amd.parallel import GradAllReducer, broadcast_module, collectives_active, init_distributed  # noqa: E402

ITERS = int(os.environ.get("HRP_RCCL_ITERS", "20"))


def main():
    rank, world, _ = init_distributed(backend="nccl")
    assert world == 1 and dist.is_initialized() and dist.get_backend() == "nccl" and collectives_active()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    B = 4
    d = {k: torch.tensor(v).to(dev) for k, v in bench.synthetic_batch(B, 808).items()}
    K = d["K"]
    kv = compute_k_values(K[:, 0, 0], K[:, 1, 1], d["bbox"])
    rot6 = rotmat_to_rot6d(d["R"])
    gmode = dict(capture_error_mode="thread_local")      # (RCCL's watchdog thread keeps polling events while this thread captures)

    def make():
        model = bench.build_model(0.0).to(dev).set_compute_dtype(torch.bfloat16).train()
        broadcast_module(model)                          # a real broadcast over the one-rank group
        params = [p for p in model.parameters() if p.requires_grad]
        opt = FusedClipAdam(params, lr=1e-4, max_norm=5.0)
        with torch.no_grad():
            kp3d, kp2d = model.robot.get_keypoints_and_projection(d["q"], rot6, d["t"], K, root=0)
            gt = dict(pose=d["q"], root_rot=model.robot.get_rotation_at_specific_root(d["q"], rot6, d["t"], root=3),
                      root_trans=kp3d[:, 3].clone(), root_uv=kp2d[:, 3].clone(), kp3d=kp3d, kp2d=kp2d,
                      mask=torch.ones(B, 7, device=dev))

        def fwd_bwd():
            loss, _ = full_loss(model(d["x_reg"], d["x_root"], kv, K), gt, K)
            loss.backward()
        return model, opt, fwd_bwd

    # ---- A: the unsplit step -------------------------------------------------------------------------------------------------
    mA, optA, fbA = make()
    mB, optB, fbB = make()
    for fb, opt in ((fbA, optA), (fbB, optB)):           # two eager steps each: plans, gradient arenas, optimizer state
        for _ in range(2):
            fb()
            opt.step()
    torch.cuda.synchronize(dev)
    (arenaA,), (arenaB,) = mA.flat_grads(), mB.flat_grads()
    pa = torch.cat([p.detach().reshape(-1) for p in mA.parameters()])
    pb = torch.cat([p.detach().reshape(-1) for p in mB.parameters()])
    assert torch.equal(pa, pb), "the two replicas diverged during the eager warm-up: the step is not bit-reproducible"
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        gA, gAu = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(gA, **gmode):
            fbA()
        with torch.cuda.graph(gAu, **gmode):
            optA.step()
        # ---- B: k + 2 graphs around the plan's cuts, collectives between them (bench.py capture_overlapped) -----------------
        sp = mB.enable_split_backward(fracs=(0.25, 0.5, 0.8, 0.9))
        assert sp is not None, "no cuts found"
        plan, groups = sp
        covered = [r for grp in groups for r in grp]
        rest = GradAllReducer.complement(covered, arenaB.numel())
        g_first, g_upd = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        g_seg = [torch.cuda.CUDAGraph() for _ in range(len(plan.bwd_cuts))]
        with torch.cuda.graph(g_first, **gmode):
            fbB()                                       # (split active: the backward stops at the first cut)
        for j, gj in enumerate(g_seg):
            with torch.cuda.graph(gj, **gmode):
                plan.run_backward(("seg", j + 1))
        with torch.cuda.graph(g_upd, **gmode):
            optB.step()
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize(dev)
    red = GradAllReducer(bucket_mb=64)
    n_coll = 0
    for it in range(ITERS):
        gA.replay()
        g_first.replay()
        w = red.start(arenaB, groups[0])
        for j, gj in enumerate(g_seg):
            gj.replay()
            if j + 1 < len(groups):
                w += red.start(arenaB, groups[j + 1])
        w += red.start(arenaB, rest)
        n_coll += len(w)
        red.finish(w, [arenaB])
        torch.cuda.synchronize(dev)
        assert torch.equal(arenaA, arenaB), f"iteration {it}: the k-cut step's reduced arena differs from the unsplit step's " \
                                            f"({((arenaA - arenaB).norm() / arenaA.norm()).item():.3e})"
        assert bool(torch.isfinite(arenaB).all())
        gAu.replay()
        g_upd.replay()
    torch.cuda.synchronize(dev)
    pa = torch.cat([p.detach().reshape(-1) for p in mA.parameters()])
    pb = torch.cat([p.detach().reshape(-1) for p in mB.parameters()])
    assert torch.equal(pa, pb), "parameters differ after the last iteration"
    print(f"rccl world-1: ok ({ITERS} iterations, {len(plan.bwd_cuts)} cuts at {plan.bwd_cuts}, {n_coll} all_reduce calls on backend "
          f"{dist.get_backend()}, arena {arenaB.numel()} elements identical to the unsplit step every iteration)")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
