#!/usr/bin/env python3
"""development tool (GPU box): the cls head's wide stride-2 convolutions (downsamp_modules, reference HRnet.py:383-405) and its final
1x1 layer through the C ABI, one forward launch at a time (tile program; 38.6 GFLOP per 3x3 layer at B = 64)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_kernels as bk  # noqa: E402
from hrpe_amd import _native as nv  # noqa: E402

DEV = bk.DEV


def mk(N, cin, cout, hw, k, s):
    dt = torch.bfloat16
    ho = hw // s
    x = torch.randn(N * hw * hw * cin, device=DEV).to(dt)
    w = torch.randn(cout, cin, k, k, device=DEV) / (cin * k * k) ** 0.5
    wp, _ = bk.pack(w, dt)
    y = torch.zeros(N * ho * ho * cout, dtype=dt, device=DEV)
    d = nv.ConvDesc()
    d.x, d.w, d.y, d.dtype = x.data_ptr(), wp.data_ptr(), y.data_ptr(), nv.HRP_BF16
    d.N, d.H, d.W, d.Cin, d.x_pitch = N, hw, hw, cin, cin
    d.Ho, d.Wo, d.Cout = ho, ho, cout
    d.y_H, d.y_W, d.y_pitch, d.res_pitch = ho, ho, cout, cout
    d.out_stride, d.in_stride = 1, s
    taps = bk.TAPS3 if k == 3 else [(0, 0)]
    d.ntaps = d.w_ntaps = len(taps)
    for i, (a, b) in enumerate(taps):
        d.dy[i], d.dx[i], d.wtap[i] = a, b, i
    d.w_cout_pad = bk.rup(cout, 32)
    return d, (x, wp, y)


if __name__ == "__main__":
    for cin, cout, hw, k, s in ((128, 256, 64, 3, 2), (256, 512, 32, 3, 2), (512, 1024, 16, 3, 2), (1024, 2048, 8, 1, 1), (256, 1024, 8, 1, 1),
                                (256, 256, 8, 3, 1), (128, 128, 16, 3, 1)):
        d, keep = mk(64, cin, cout, hw, k, s)
        t = bk.timeit(lambda: nv.call("hrp_conv2d_fwd", C.byref(d), None))
        ho = hw // s
        fl = 2.0 * 64 * ho * ho * cout * cin * k * k
        by = 2.0 * 64 * (hw * hw * cin + ho * ho * cout) + 2.0 * cout * cin * k * k
        print(f"{cin:4d}>{cout:4d} k{k} s{s} @{hw:3d}: {t:7.1f} us  {fl / t / 1e6:6.0f} TFLOP/s  {by / t / 1e3:6.0f} GB/s")
