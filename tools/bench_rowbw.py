#!/usr/bin/env python3
"""Micro-benchmark of the fused row-strip backward launch (hrp_rowbw_*, csrc/conv_rowbw.hip) against the launches it replaces
(the batched data gradient with its side output + the batched weight gradient) on the two high-resolution branches of one
HRNet-W32 trunk (development tool; run on the GPU box).

    python tools/bench_rowbw.py [--batch 64] [--wgs 0,128,192]
"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_batch as bb  # noqa: E402
import bench_kernels as bk  # noqa: E402
from hrpe_amd import _native as nv  # noqa: E402

DEV = bk.DEV


def problems(N, kind):
    bb.KIND = kind
    out = []
    for c, hw in bb.CLASSES[:2]:
        d, bufs = bb.mk_conv(N, hw, c, torch.bfloat16)
        d.pro_side2 = None        # (the plan's default: the shortcut gradient travels as a masked residual of conv1's data gradient)
        out.append((d, bufs, c, hw))
    return out


def wgrad_descs(N, probs, keep):
    descs = []
    for d, bufs, c, hw in probs:
        g = nv.WgradDesc()
        dw = torch.zeros(c, c, 9, device=DEV)
        xop = torch.randn(N * hw * hw * c, device=DEV).to(torch.bfloat16)
        g.x, g.dy, g.dw, g.dtype = xop.data_ptr(), d.pro_side, dw.data_ptr(), d.dtype
        g.N, g.H, g.W, g.Cin, g.x_pitch = N, hw, hw, c, c
        g.Ho, g.Wo, g.Cout, g.dy_pitch = hw, hw, c, c
        g.in_stride, g.ntaps = 1, 9
        for i, (a, b) in enumerate(bk.TAPS3):
            g.dy_t[i], g.dx_t[i] = a, b
        g.dw_cin, g.accumulate, g.phase = c, 1, 1
        descs.append(g)
        keep += [dw, xop]
    arr = (nv.WgradDesc * len(descs))(*descs)
    info = nv.BatchInfo()
    nv.check(nv.lib().hrp_batch_prepare(nv.BATCH_WGRAD, arr, len(descs), None, C.byref(info)), "query")
    for i, g in enumerate(descs):
        ws = torch.zeros(info.ws_bytes[i] // 4 + 4, device=DEV)
        g.workspace, g.workspace_bytes = ws.data_ptr(), ws.numel() * 4
        keep.append(ws)
    return descs


def fused(N, probs, keep, wgs, act):
    qs = []
    for d, bufs, c, hw in probs:
        q = nv.RowBwDesc()
        C.memmove(C.byref(q.conv), C.byref(d), C.sizeof(nv.ConvDesc))
        q.conv.pro_side = None
        xop = torch.randn(N * hw * hw * c, device=DEV).to(torch.bfloat16)
        dw = torch.zeros(c * c * 9, device=DEV)
        q.wg_x, q.dw, q.wg_act, q.accumulate = (d.bnb_x if (act and d.bnb_x) else xop.data_ptr()), dw.data_ptr(), 1 if (act and d.bnb_x) else 0, 1
        keep += [xop, dw]
        qs.append(q)
    n = len(qs)
    arr = (nv.RowBwDesc * n)(*qs)
    info = nv.RowBwInfo()
    L = nv.lib()
    nv.check(L.hrp_rowbw_prepare(arr, n, wgs, None, C.byref(info)), "query")
    tot_ws = 0
    for i in range(n):
        ws = torch.zeros(int(info.ws_bytes[i]) // 4 + 4, device=DEV)
        arr[i].workspace, arr[i].workspace_bytes = ws.data_ptr(), int(info.ws_bytes[i])
        keep.append(ws)
        tot_ws += int(info.ws_bytes[i])
    table = (C.c_char * int(L.hrp_rowbw_table_bytes()))()
    nv.check(L.hrp_rowbw_prepare(arr, n, wgs, table, C.byref(info)), "prepare")
    folds = (nv.WgradFoldDesc * n)()
    nv.check(L.hrp_rowbw_fold_descs(arr, C.byref(info), folds), "folds")
    finfo = nv.BatchInfo()
    fhost = (C.c_char * int(L.hrp_batch_table_bytes(nv.BATCH_WGRAD_FOLD, n)))()
    nv.check(L.hrp_batch_prepare(nv.BATCH_WGRAD_FOLD, folds, n, fhost, C.byref(finfo)), "fold prepare")
    ftab = torch.frombuffer(bytearray(bytes(fhost)), dtype=torch.uint8).to(DEV)
    keep += [table, arr, ftab, finfo, info]
    run = lambda: nv.check(L.hrp_rowbw_launch(table, C.byref(info), None), "launch")           # noqa: E731
    runf = lambda: nv.check(L.hrp_batch_launch(ftab.data_ptr(), C.byref(finfo), None), "fold")    # noqa: E731
    return run, runf, info, tot_ws


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--wgs", default="0,128,192")
    a = ap.parse_args()
    N = a.batch
    for kind in ("g2e", "g1e"):
        keep = []
        probs = problems(N, kind)
        tens = sum(N * hw * hw * c * 2 for _, _, c, hw in probs)
        conv = bb.Batch("conv", [p[0] for p in probs])
        t_conv = bk.timeit(conv)
        wg = bb.Batch("wgrad", wgrad_descs(N, probs, keep))
        t_wg = bk.timeit(wg)
        print(f"{kind}: separate  data gradient {t_conv:6.1f} us + weight gradient {t_wg:6.1f} us = {t_conv + t_wg:6.1f} us   (one tensor pass = {tens / 1e6:.1f} MB)")
        for wgs in [int(v) for v in a.wgs.split(",")]:
            run, runf, info, tot_ws = fused(N, probs, keep, wgs, act=(kind == "g2e"))
            t_f, t_fold = bk.timeit(run), bk.timeit(runf)
            if bk.TIMELINE:
                L = nv.lib()
                torch.cuda.synchronize()
                form = L.hrp_debug_rowbw_form(C.byref(keep[-4][0]))
                L.hrp_debug_rowbw_timeline(None, 0, 1, form)
                run()
                torch.cuda.synchronize()
                host = torch.zeros(1024 * 32, dtype=torch.int64)
                L.hrp_debug_rowbw_timeline(C.c_void_p(host.data_ptr()), 1024, 0, form)
                tl4 = host.view(-1, 4, 8).double()[:info.grid]
                tl = tl4[:, :2]
                names = ["wait dma", "B2 wait", "wgrad loop", "stage issue", "act X8", "B1 wait", "-"]
                t0 = tl[:, 0, 0].min()
                for lo, hi, nm in ((0, info.first_wg[1] if info.n > 1 else info.grid, "C=32 workgroups"), (info.first_wg[1] + 1 if info.n > 1 else info.grid, info.grid, "C=64 workgroups")):
                    if lo >= hi:
                        continue
                    for itx in range(2):
                        t = tl[lo:hi, itx]
                        t = t[t[:, 6] > 0]
                        if t.shape[0] == 0:
                            continue
                        print(f"        {nm} strip {itx}: start {((t[:, 0] - t0).mean() / 100):6.2f}  " +
                              "  ".join(f"{names[k]} {((t[:, k + 1] - t[:, k]).mean() / 100):.2f}" for k in range(6)) +
                              f"  total {((t[:, 6] - t[:, 0]).mean() / 100):.2f} us")
                        ta = tl4[lo:hi, 2 + itx]
                        ta = ta[ta[:, 7] > 0]
                        an = ["dgrad loop", "issue loads", "B2 wait", "epilogue", "stat reduce", "prologue", "B1 wait"]
                        if ta.shape[0]:
                            print(f"        {'':15s} role 0 {itx}: start {((ta[:, 0] - t0).mean() / 100):6.2f}  " +
                                  "  ".join(f"{an[k]} {((ta[:, k + 1] - ta[:, k]).mean() / 100):.2f}" for k in range(7)) +
                                  f"  total {((ta[:, 7] - ta[:, 0]).mean() / 100):.2f} us")
            # bytes of the fused launch: staged operand 1.25 + second prologue operand 1.25 + X 1 + output 1 (+ g2e: mask bits;
            # g1e: residual 1 + epilogue-reduce operand 1) + slabs
            passes = 4.5 if kind == "g2e" else 6.5
            by = passes * tens + tot_ws
            print(f"      fused (grid {info.grid:3d}, slabs {tot_ws / 1e6:5.1f} MB): {t_f:6.1f} us  ({by / t_f / 1e6:5.2f} TB/s on {by / 1e6:.0f} MB)"
                  f" + fold {t_fold:5.1f} us")
