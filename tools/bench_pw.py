#!/usr/bin/env python3
"""Micro-benchmark of the pointwise convolution kernel (csrc/conv_pw.h) against the general tile program on the dense 1x1
layers of the benchmark step (development tool; run on the GPU box)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_kernels as bk  # noqa: E402
from hrpe_amd import _native as nv  # noqa: E402

DEV = bk.DEV


def mk(N, hw, cin, cout, kind):
    M = N * hw * hw
    x = torch.randn(M * cin, device=DEV).to(torch.bfloat16)
    w = torch.randn(cout, cin, 1, 1, device=DEV) / cin ** 0.5
    wp, _ = bk.pack(w, torch.bfloat16)
    y = torch.zeros(M * cout, dtype=torch.bfloat16, device=DEV)
    st = torch.zeros(16 * cout, dtype=torch.float64, device=DEV)
    d = nv.ConvDesc()
    d.x, d.w, d.y, d.dtype = x.data_ptr(), wp.data_ptr(), y.data_ptr(), nv.HRP_BF16
    d.N, d.H, d.W, d.Cin, d.x_pitch = N, hw, hw, cin, cin
    d.Ho, d.Wo, d.Cout = hw, hw, cout
    d.y_H, d.y_W, d.y_pitch, d.res_pitch = hw, hw, cout, cout
    d.out_stride, d.in_stride, d.ntaps, d.w_ntaps, d.w_cout_pad = 1, 1, 1, 1, bk.rup(cout, 32)
    keep = [x, wp, y, st]
    by = M * (cin + cout) * 2
    if kind == "S":
        d.stats = st.data_ptr()
    elif kind == "R":
        d.res = y.data_ptr()
        by += M * cout * 2
    elif kind == "Sbm":
        bx = torch.randn(M * cout, device=DEV).to(torch.bfloat16)
        mk_ = torch.randint(0, 255, (M * cout // 8,), dtype=torch.uint8, device=DEV)
        cs = torch.rand(2 * cout, device=DEV) + 0.5
        keep += [bx, mk_, cs]
        d.stats, d.bnb_x, d.bnb_x_pitch, d.bnb_mask, d.bnb_mask_pitch, d.bnb_consts = st.data_ptr(), bx.data_ptr(), cout, mk_.data_ptr(), cout // 8, cs.data_ptr()
        by += M * cout * 2 + M * cout // 8
    return d, keep, by


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    for cin, cout, hw, kind in [(64, 256, 64, "S"), (256, 64, 64, "S"), (256, 64, 64, "Sbm"), (64, 256, 64, "R"), (256, 64, 64, ""),
                                (32, 128, 64, "S"), (128, 32, 64, "Sbm"), (64, 64, 64, "S"), (32, 32, 64, "S"), (64, 256, 32, "S"),
                                (256, 64, 32, "Sbm"), (64, 32, 32, "S"), (32, 64, 32, "R"), (128, 512, 16, "S"), (128, 64, 16, "S")]:
        d, keep, by = mk(B, hw, cin, cout, kind)
        os.environ["HRP_PW_MIN_PIXELS"] = "1"
        if nv.lib().hrp_conv_pointwise(C.byref(d)) != 1:      # (e.g. Cin = 256 with the epilogue reduce: stays on the tile program)
            t_tile = bk.timeit(lambda: nv.call("hrp_conv2d_fwd", C.byref(d), None))
            print(f"{cin:4d} -> {cout:4d} @ {hw:2d}x{hw:2d} x{B} [{kind:3s}]  not a pointwise problem            tile {t_tile:7.1f} us {by / t_tile / 1e3:6.0f} GB/s")
            continue
        t_pw = bk.timeit(lambda: nv.call("hrp_conv2d_fwd", C.byref(d), None))
        os.environ["HRP_PW_MIN_PIXELS"] = str(2 ** 31 - 1)
        t_tile = bk.timeit(lambda: nv.call("hrp_conv2d_fwd", C.byref(d), None))
        os.environ["HRP_PW_MIN_PIXELS"] = "1"
        print(f"{cin:4d} -> {cout:4d} @ {hw:2d}x{hw:2d} x{B} [{kind:3s}]  pointwise {t_pw:7.1f} us {by / t_pw / 1e3:6.0f} GB/s   tile {t_tile:7.1f} us {by / t_tile / 1e3:6.0f} GB/s")
