"""dev aid: BasicBlock fp32 train step repeated with fresh modules; on a dx mismatch print where it is"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_kernels import _rand_sd, _load_into, DEV
from hrpe_amd.lib.models.backbones import HRnet as H
torch.manual_seed(0)
x = torch.randn(16, 32, 64, 64, generator=torch.Generator().manual_seed(5))
gy = torch.randn(16, 32, 64, 64, generator=torch.Generator().manual_seed(6))
ref = None
bad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    m = H.BasicBlock(32, 32)
    _load_into(m, _rand_sd(m, 3))
    m = m.to(DEV).set_compute_dtype(torch.float32).train()
    xd = x.to(DEV).requires_grad_(True)
    y = m(xd)
    torch.cuda.synchronize()
    yv = y.detach().clone()
    (y * gy.to(DEV)).sum().backward()
    torch.cuda.synchronize()
    g = xd.grad.detach().clone()
    bufs = {k: b.detach().clone() for k, b in m.named_buffers()}
    plan = [r.plan for r in m._plans.values()][-1]
    snap = [t.detach().clone() for t in plan.keep if torch.is_tensor(t)]
    grads = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
    if ref is None:
        ref, refg, refy, refb, refsnap = g, grads, yv, bufs, snap
        continue
    dy = (yv - refy).abs()
    if float(dy.max()) > 0:
        idx = (dy > 0).nonzero()
        print(f"iter {it}: FORWARD differs: {int((dy > 0).sum())} elems, max {float(dy.max()):.3e}; n {idx[:,0].unique().tolist()} y {idx[:,2].unique().tolist()[:20]} x {idx[:,3].unique().tolist()[:20]}")
    for k in bufs:
        if bufs[k].dtype.is_floating_point and float((bufs[k] - refb[k]).abs().max()) > 0:
            print(f"iter {it}: buffer {k} differs by {float((bufs[k] - refb[k]).abs().max()):.3e}")
    e = float((g - ref).norm() / ref.norm())
    if e > 1e-4:
        bad += 1
        for bi, (a, b) in enumerate(zip(snap, refsnap)):
            if a.shape != b.shape or not a.dtype.is_floating_point and a.dtype != torch.uint8:
                continue
            af, bf = a.float(), b.float()
            dd = (af - bf).abs() > 1e-4 * max(float(bf.abs().max()), 1e-20)
            if bool(dd.any()):
                ii = dd.nonzero().flatten()
                print(f"   keep[{bi}] {a.dtype} n={a.numel()} differs at {int(dd.sum())} elems, first {int(ii[0])} last {int(ii[-1])}, maxdiff {float((af-bf).abs().max()):.3e} refmax {float(bf.abs().max()):.3e}")
        d = (g - ref).abs() > 1e-4 * float(ref.abs().max())
        idx = d.nonzero()
        print(f"iter {it}: dx rel {e:.5f}; bad elems {int(d.sum())} of {d.numel()}; n {idx[:,0].unique().tolist()} c {idx[:,1].unique().tolist()[:40]} "
              f"y {idx[:,2].unique().tolist()[:70]} x {idx[:,3].unique().tolist()[:70]}")
        for k in grads:
            ee = float((grads[k] - refg[k]).norm() / (refg[k].norm() + 1e-30))
            if ee > 1e-4:
                print("   param grad differs:", k, ee)
    del m
print("bad", bad)
