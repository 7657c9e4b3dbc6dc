#!/usr/bin/env python3
"""Micro-benchmark of the batched launches (hrp_batch_*) on the lock-step layers of an HRNet-W32 stage
(development tool; run on the GPU box).

    python tools/bench_batch.py [conv|wgrad|all] [--batch 64] [--nets 2] [--branches 4]

HRP_TIMELINE=1 (needs `make -C .../csrc timeline`): per-problem phase means of the conv workgroups.
"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_kernels as bk  # noqa: E402
from hrpe_amd import _native as nv  # noqa: E402

DEV = bk.DEV
CLASSES = [(32, 64), (64, 32), (128, 16), (256, 8)]
KIND = "plain"
X3 = False


def mk_conv(N, hw, c, dtype, stats=True, k=3):
    esz = 2
    x = torch.randn(N * hw * hw * c, device=DEV).to(dtype)
    w = torch.randn(c, c, k, k, device=DEV) / (c * k * k) ** 0.5
    wp, _ = bk.pack(w, dtype)
    y = torch.zeros(N * hw * hw * c, dtype=dtype, device=DEV)
    st = torch.zeros(16 * c, dtype=torch.float64, device=DEV)
    d = nv.ConvDesc()
    d.x, d.w, d.y = x.data_ptr(), wp.data_ptr(), y.data_ptr()
    d.dtype = nv.HRP_BF16
    d.N, d.H, d.W, d.Cin, d.x_pitch = N, hw, hw, c, c
    d.Ho, d.Wo, d.Cout = hw, hw, c
    d.y_H, d.y_W, d.y_pitch, d.res_pitch = hw, hw, c, c
    d.out_stride, d.in_stride = 1, 1
    taps = bk.TAPS3 if k == 3 else [(0, 0)]
    d.ntaps = d.w_ntaps = len(taps)
    for i, (a, b) in enumerate(taps):
        d.dy[i], d.dx[i], d.wtap[i] = a, b, i
    d.w_cout_pad = bk.rup(c, 32)
    if stats:
        d.stats = st.data_ptr()
    bufs = [x, wp, y, st]
    kind = KIND
    if kind != "plain" and k == 3:
        # the launch kinds of a fused BasicBlock (plan.conv_bn_relu_conv): "pro1" conv2 forward, "g2" conv2 data gradient
        # (epilogue reduce), "g1" conv1 data gradient (apply prologue + residual), "g2e" / "g1e" with the block-end backward
        n = N * hw * hw * c
        t = lambda: torch.randn(n, device=DEV).to(dtype)     # noqa: E731
        sts = torch.rand(16 * c, dtype=torch.float64, device=DEV) * 100 + 50
        sts[8 * c:] += 1e5
        gam, bet = torch.ones(c, device=DEV), torch.zeros(c, device=DEV)
        bs = torch.zeros(16 * c, dtype=torch.float64, device=DEV)
        x2, side, side2, x3 = t(), t(), t(), t()
        mask = torch.randint(0, 255, (n // 8,), dtype=torch.uint8, device=DEV)
        bufs += [sts, gam, bet, bs, x2, side, side2, x3, mask]
        cnt = float(N * hw * hw)
        if kind == "pro1":
            d.pro_mode, d.pro_stats, d.pro_gamma, d.pro_beta, d.pro_count, d.pro_eps = 1, sts.data_ptr(), gam.data_ptr(), bet.data_ptr(), cnt, 1e-5
            d.pro_side = side.data_ptr()
        if kind in ("g2", "g2e"):
            d.stats, d.bnb_x, d.bnb_x_pitch = bs.data_ptr(), x2.data_ptr(), c
            d.bnb_stats, d.bnb_gamma, d.bnb_beta, d.bnb_count, d.bnb_eps = sts.data_ptr(), gam.data_ptr(), bet.data_ptr(), cnt, 1e-5
        if kind in ("g1", "g1e", "g2e"):
            d.pro_mode, d.pro_stats, d.pro_gamma, d.pro_beta, d.pro_count, d.pro_eps = 2, sts.data_ptr(), gam.data_ptr(), bet.data_ptr(), cnt, 1e-5
            d.pro_x2, d.pro_bsums, d.pro_side = x3.data_ptr(), bs.data_ptr(), side.data_ptr()
        if kind in ("g1", "g1e"):
            d.res, d.stats = y.data_ptr(), 0
        if kind == "g1e":
            d.stats, d.bnb_x, d.bnb_x_pitch, d.bnb_mask, d.bnb_mask_pitch = bs.data_ptr(), x2.data_ptr(), c, mask.data_ptr(), c // 8
            d.bnb_stats, d.bnb_gamma, d.bnb_beta, d.bnb_count, d.bnb_eps = sts.data_ptr(), gam.data_ptr(), bet.data_ptr(), cnt, 1e-5
        if kind == "g2e":
            d.pro_mask, d.pro_side2 = mask.data_ptr(), side2.data_ptr()
    return d, tuple(bufs)


class Batch:
    def __init__(self, fam, descs):
        famid = {"conv": nv.BATCH_CONV, "wgrad": nv.BATCH_WGRAD}[fam]
        n = len(descs)
        self.fam, self.items = fam, descs
        arr = (type(descs[0]) * n)(*descs)
        self.info = nv.BatchInfo()
        nb = int(nv.lib().hrp_batch_table_bytes(famid, n))
        host = (C.c_char * nb)()
        nv.check(nv.lib().hrp_batch_prepare(famid, arr, n, host, C.byref(self.info)), "prepare")
        self.table = torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8).to(DEV)

    def __call__(self):
        nv.check(nv.lib().hrp_batch_launch(self.table.data_ptr(), C.byref(self.info), None), "launch")


def conv_batch(N, nets, branches, dtype):
    keep, descs = [], []
    for _ in range(nets):
        for c, hw in CLASSES[:branches]:
            d, bufs = mk_conv(N, hw, c, dtype)
            descs.append(d)
            keep.append(bufs)
    singles = [bk.timeit(lambda d=d: nv.call("hrp_conv2d_fwd", C.byref(d), None)) for d in descs[:branches]]
    b = Batch("conv", descs)
    us = bk.timeit(b)
    fl = sum(2.0 * d.N * d.Ho * d.Wo * d.Cout * d.Cin * 9 for d in descs)
    by = sum(2.0 * d.N * d.Ho * d.Wo * d.Cout * 2 for d in descs)
    print(f"conv batch{len(descs)} (nets {nets} x branches {branches}, B={N}): {us:7.1f} us  {fl / us / 1e6:6.1f} TFLOP/s  "
          f"{by / us / 1e3:6.0f} GB/s  grid {b.info.grid} lds {b.info.lds_bytes}  | one by one: "
          + " ".join(f"{t:.1f}" for t in singles) + f" (sum x nets {sum(singles) * nets:.1f})")
    if bk.TIMELINE:
        L = nv.lib()
        torch.cuda.synchronize()
        L.hrp_debug_conv_timeline_batch(None, 0, 1)
        b()
        torch.cuda.synchronize()
        host = torch.zeros(8192 * 8, dtype=torch.int64)
        L.hrp_debug_conv_timeline_batch(C.c_void_p(host.data_ptr()), 8192, 0)
        tl = host.view(-1, 8).double()
        t0 = tl[tl[:, 0] > 0][:, 0].min()
        names = ["entry", "setup", "issued", "stage0", "loop end", "in lds", "stored", "end"]
        for i in range(b.info.n):
            lo, hi = b.info.blk0[i], min(b.info.blk0[i + 1], 8192)
            if lo >= hi:
                continue
            t = tl[lo:hi]
            t = t[t[:, 0] > 0]
            print(f"   problem {i}: blocks {lo}..{hi}  start {((t[:, 0] - t0).mean() / 100):6.2f}/{((t[:, 0] - t0).max() / 100):6.2f}  "
                  f"end {((t[:, 7] - t0).mean() / 100):6.2f}/{((t[:, 7] - t0).max() / 100):6.2f}  phases: "
                  + "  ".join(f"{names[k + 1]} {((t[:, k + 1] - t[:, k]).mean() / 100):.2f}" for k in range(7))
                  + f"  total {((t[:, 7] - t[:, 0]).mean() / 100):.2f}")
    return keep


def wgrad_batch(N, nets, branches, dtype):
    keep, descs = [], []
    for _ in range(nets):
        for c, hw in CLASSES[:branches]:
            d, bufs = mk_conv(N, hw, c, dtype, stats=False)
            g = nv.WgradDesc()
            dw = torch.zeros(c, c, 9, device=DEV)
            g.x, g.dy, g.dw, g.dtype = d.x, d.y, dw.data_ptr(), d.dtype
            if X3:      # fp32 tensors, three bf16 products (HRP_F32X3)
                xf, yf = torch.randn(N * hw * hw * c, device=DEV), torch.randn(N * hw * hw * c, device=DEV)
                bufs = bufs + (xf, yf)
                g.x, g.dy, g.dtype = xf.data_ptr(), yf.data_ptr(), nv.HRP_F32X3
            g.N, g.H, g.W, g.Cin, g.x_pitch = N, hw, hw, c, c
            g.Ho, g.Wo, g.Cout, g.dy_pitch = hw, hw, c, c
            g.in_stride, g.ntaps = 1, 9
            for i, (a, b) in enumerate(bk.TAPS3):
                g.dy_t[i], g.dx_t[i] = a, b
            g.dw_cin, g.accumulate = c, 1
            descs.append(g)
            keep.append((bufs, dw))
    arr = (nv.WgradDesc * len(descs))(*descs)
    info = nv.BatchInfo()
    nv.check(nv.lib().hrp_batch_prepare(nv.BATCH_WGRAD, arr, len(descs), None, C.byref(info)), "query")
    wss = []
    for i, g in enumerate(descs):
        ws = torch.zeros(info.ws_bytes[i] // 4 + 4 + ((1 << 18) if bk.TIMELINE else 0), device=DEV)
        g.workspace, g.workspace_bytes = ws.data_ptr(), ws.numel() * 4
        keep.append(ws)
        wss.append(ws)
    singles = []
    for g in descs[:branches]:
        g1 = nv.WgradDesc.from_buffer_copy(bytes(g))
        nb = int(nv.lib().hrp_wgrad_workspace_bytes(C.byref(g1)))
        ws = torch.zeros(nb // 4 + 4, device=DEV)
        g1.workspace, g1.workspace_bytes = ws.data_ptr(), ws.numel() * 4
        singles.append(bk.timeit(lambda g1=g1: nv.call("hrp_conv2d_bwd_weight", C.byref(g1), None)))
    b = Batch("wgrad", descs)
    us = bk.timeit(b)
    fl = sum(2.0 * d.N * d.Ho * d.Wo * d.Cout * d.Cin * 9 for d in descs)
    if bk.TIMELINE:
        for ws in wss:
            ws.view(torch.int64)[-(1 << 17):].zero_()
        b()
        torch.cuda.synchronize()
        names = ["entry", "setup", "issued", "t0 ready", "t0 done", "t1 ready", "t1 done", "t2 ready", "t2 done",
                 "t3 ready", "t3 done", "loop end", "round1", "stored"]
        tls = [ws.view(torch.int64)[-(1 << 17):].cpu().view(-1, 16).double() for ws in wss]
        t0 = min(float(t[t[:, 0] > 0][:, 0].min()) for t in tls)
        for i, t in enumerate(tls[:branches]):
            t = t[t[:, 0] > 0]
            print(f"   problem {i} ({t.shape[0]} workgroups; us after the first entry, mean/max): " +
                  "  ".join(f"{n} {((t[:, k] - t0).mean() / 100):.2f}/{((t[:, k] - t0).max() / 100):.2f}" for k, n in enumerate(names) if t[:, k].max() > 0))
    print(f"wgrad batch{len(descs)} (nets {nets} x branches {branches}, B={N}): {us:7.1f} us  {fl / us / 1e6:6.1f} TFLOP/s  "
          f"grid {b.info.grid}+{b.info.grid3} (eight-wave)+{b.info.grid2} lds {b.info.lds_bytes}/{b.info.lds_bytes3}  | one by one: " + " ".join(f"{t:.1f}" for t in singles)
          + f" (sum x nets {sum(singles) * nets:.1f})")
    return keep


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="?", default="all")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--nets", type=int, default=2)
    ap.add_argument("--branches", type=int, default=4)
    ap.add_argument("--kind", default="plain", choices=["plain", "pro1", "g2", "g1", "g2e", "g1e"])
    ap.add_argument("--x3", action="store_true", help="weight gradients in the fp32x3 mode")
    a = ap.parse_args()
    KIND = a.kind
    X3 = a.x3
    dt = torch.bfloat16
    if a.what in ("conv", "all"):
        conv_batch(a.batch, a.nets, a.branches, dt)
        conv_batch(a.batch, 1, a.branches, dt)
        for i in range(4):
            CLS = CLASSES
            CLASSES = [CLS[i]]
            conv_batch(a.batch, 2, 1, dt)
            CLASSES = CLS
    if a.what in ("wgrad", "all"):
        wgrad_batch(a.batch, a.nets, a.branches, dt)
        wgrad_batch(a.batch, 1, a.branches, dt)
        for i in range(4):
            CLS = CLASSES
            CLASSES = [CLS[i]]
            wgrad_batch(a.batch, 2, 1, dt)
            CLASSES = CLS
