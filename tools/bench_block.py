#!/usr/bin/env python3
"""Micro-benchmark of the fused inference BasicBlock launch (hrp_block_*, csrc/conv_block.h) against the two batched conv
launches it replaces, on the two high-resolution branches of one HRNet-W32 trunk (development tool; run on the GPU box).

    python tools/bench_block.py [--batch 64]
"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import bench_kernels as bk  # noqa: E402
from hrpe_amd import _native as nv  # noqa: E402
import test_gpu_block as tb  # noqa: E402

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    a = ap.parse_args()
    N = a.batch
    keep = []
    b32, _, (a1, a2, _), _, _ = tb.make_block(nv, 32, N, 64, 1, keep)
    b64, _, (c1, c2, _), _, _ = tb.make_block(nv, 64, N, 32, 2, keep)
    L = nv.lib()
    for name, blocks in (("C=32", [b32]), ("C=64", [b64]), ("C=32 + C=64", [b32, b64])):
        n = len(blocks)
        arr = (nv.BlockDesc * n)(*blocks)
        info = nv.BlockInfo()
        table = (C.c_char * int(L.hrp_block_table_bytes()))()
        nv.check(L.hrp_block_prepare(arr, n, table, C.byref(info)), "prepare")
        t = bk.timeit(lambda: nv.check(L.hrp_block_launch(table, C.byref(info), None), "launch"))
        pix = lambda b: N * b.conv1.H * b.conv1.W     # noqa: E731
        tens = sum(pix(b) * b.conv1.Cin * 2 for b in blocks)
        flops = sum(2 * 2 * 9 * b.conv1.Cin ** 2 * pix(b) for b in blocks)
        if bk.TIMELINE:
            L.hrp_debug_block_timeline(None, 1)
            nv.check(L.hrp_block_launch(table, C.byref(info), None), "launch")
            torch.cuda.synchronize()
            host = torch.zeros(256 * 2 * 8, dtype=torch.int64)
            L.hrp_debug_block_timeline(C.c_void_p(host.data_ptr()), 0)
            tl = host.view(256, 2, 8).double()[:min(info.grid, 256)]
            for lo, hi, nm in ((0, info.first_wg[1] if info.n > 1 else info.grid, f"C={blocks[0].conv1.Cin}"), (info.first_wg[1] if info.n > 1 else info.grid, info.grid, "C=64")):
                if lo >= hi:
                    continue
                for role, names in ((0, ["stage", "mfma loop", "epilogue", "vmcnt", "barrier"]), (1, ["epilogue", "mfma loop", "-", "-", "barrier"])):
                    t_ = tl[lo:hi, role]
                    t_ = t_[t_[:, 5] > 0]
                    if t_.shape[0]:
                        print(f"        {nm} role {role} iteration 6 (cycles): " + "  ".join(f"{names[k]} {(t_[:, k + 1] - t_[:, k]).mean():.0f}" for k in range(5)) + f"  total {(t_[:, 5] - t_[:, 0]).mean():.0f}")
        print(f"{name:12s} fused block: {t:6.1f} us  grid {info.grid:4d}  {2 * tens / t / 1e6:5.2f} TB/s (x + out = {2 * tens / 1e6:.0f} MB)  "
              f"{flops / t / 1e6:6.1f} TFLOP/s")
    for name, ds in (("C=32", [a1, a2]), ("C=64", [c1, c2])):
        ts = [bk.timeit(lambda d=d: nv.call("hrp_conv2d_fwd", C.byref(d), None)) for d in ds]
        print(f"{name:12s} unfused: conv1 {ts[0]:6.1f} us + conv2 {ts[1]:6.1f} us = {sum(ts):6.1f} us")
