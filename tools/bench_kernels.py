#!/usr/bin/env python3
"""Per-kernel micro-benchmark through the C ABI (development tool; run on the GPU box).

    python tools/bench_kernels.py [conv|wgrad|ew|all] [--batch 64]

Shapes are the dominant conv classes of HRNet-W32 at 256x256.  Each kernel is
launched `reps` times back to back on one stream and timed with HIP events around the whole train, so the
numbers are pure device time per launch (no host gaps).
"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hrpe_amd  # noqa: E402,F401
from hrpe_amd import _native as nv  # noqa: E402

TIMELINE = bool(os.environ.get("HRP_TIMELINE"))   # needs `make -C .../csrc timeline`
if TIMELINE:
    nv.LIB_PATH = nv.LIB_PATH.replace("libhrp_hip.so", "libhrp_hip_tl.so")
DEV = torch.device("cuda:0")
TAPS3 = [(ky - 1, kx - 1) for ky in range(3) for kx in range(3)]


def rup(a, b):
    return (a + b - 1) // b * b


def pack(w, dtype):
    cout, cin = w.shape[0], w.shape[1]
    ntaps = w.shape[2] * w.shape[3]
    esz = 2 if dtype == torch.bfloat16 else 4
    ck = 32 // esz
    nf = -(-cin // ck) * ntaps * rup(cout, 32) * ck
    nb = -(-cout // ck) * ntaps * rup(cin, 32) * ck
    dst = torch.zeros(nf, dtype=dtype, device=DEV)
    dst_t = torch.zeros(nb, dtype=dtype, device=DEV)
    tab = (nv.PackEntry * 1)()
    tab[0].src, tab[0].dst, tab[0].dst_t = w.data_ptr(), dst.data_ptr(), dst_t.data_ptr()
    tab[0].Cout, tab[0].Cin, tab[0].ntaps = cout, cin, ntaps
    tdev = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).to(DEV)
    nv.call("hrp_pack_weights", tdev.data_ptr(), 1, nv.HRP_BF16 if esz == 2 else nv.HRP_F32, max(nf, nb), None)
    torch.cuda.synchronize()
    return dst, dst_t


def timeit(fn, reps=int(os.environ.get("HRP_REPS", "50"))):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3  # us


def conv_case(N, H, W, cin, cout, k, stride, dtype, stats):
    esz = 2 if dtype == torch.bfloat16 else 4
    Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
    x = torch.randn(N * H * W * rup(cin, 8), device=DEV).to(dtype)
    w = torch.randn(cout, cin, k, k, device=DEV) / (cin * k * k) ** 0.5
    wp, wpt = pack(w, dtype)
    y = torch.zeros(N * Ho * Wo * rup(cout, 8), dtype=dtype, device=DEV)
    st = torch.zeros(16 * cout, dtype=torch.float64, device=DEV)
    d = nv.ConvDesc()
    d.x, d.w, d.y = x.data_ptr(), wp.data_ptr(), y.data_ptr()
    d.dtype = nv.HRP_BF16 if esz == 2 else nv.HRP_F32
    d.N, d.H, d.W, d.Cin, d.x_pitch = N, H, W, rup(cin, 8), rup(cin, 8)
    d.Ho, d.Wo, d.Cout = Ho, Wo, cout
    d.y_H, d.y_W, d.y_pitch, d.res_pitch = Ho, Wo, rup(cout, 8), rup(cout, 8)
    d.out_stride, d.in_stride = 1, stride
    taps = TAPS3 if k == 3 else [(0, 0)]
    d.ntaps = d.w_ntaps = len(taps)
    for i, (a, b) in enumerate(taps):
        d.dy[i], d.dx[i], d.wtap[i] = a, b, i
    d.w_cout_pad = rup(cout, 32)
    if stats:
        d.stats = st.data_ptr()
    us = timeit(lambda: nv.call("hrp_conv2d_fwd", C.byref(d), None))
    if TIMELINE:
        L = nv.lib()
        torch.cuda.synchronize()
        L.hrp_debug_conv_timeline(None, 0, 1)
        nv.call("hrp_conv2d_fwd", C.byref(d), None)
        torch.cuda.synchronize()
        host = torch.zeros(8192 * 8, dtype=torch.int64)
        L.hrp_debug_conv_timeline(C.c_void_p(host.data_ptr()), 8192, 0)
        tl = host.view(-1, 8)
        tl = tl[tl[:, 0] > 0].double()
        t0 = tl[:, 0].min()
        names = ["entry", "setup", "issued", "stage0", "loop end", "in lds", "stored", "end"]
        print(f"      conv timeline over {tl.shape[0]} workgroups (us after the first entry; mean/max): " +
              "  ".join(f"{n} {((tl[:, i] - t0).mean() / 100):.2f}/{((tl[:, i] - t0).max() / 100):.2f}"
                        for i, n in enumerate(names)))
        print("      per-workgroup phase means (us): " +
              "  ".join(f"{names[i + 1]} {((tl[:, i + 1] - tl[:, i]).mean() / 100):.2f}" for i in range(7)))
    fl = 2.0 * N * Ho * Wo * cout * cin * len(taps)
    by = (x.numel() + y.numel()) * esz
    print(f"conv  N={N} {cin:4d}->{cout:4d} k{k} s{stride} @{H:3d}x{W:<3d} stats={int(stats)}: {us:8.1f} us  "
          f"{fl / us / 1e6:7.1f} TFLOP/s  {by / us / 1e3:7.1f} GB/s")
    # weight gradient of the same layer
    g = nv.WgradDesc()
    dw = torch.zeros(cout, cin, k * k, device=DEV)
    g.x, g.dy, g.dw = x.data_ptr(), y.data_ptr(), dw.data_ptr()
    g.dtype = d.dtype
    g.N, g.H, g.W, g.Cin, g.x_pitch = N, H, W, rup(cin, 8), rup(cin, 8)
    g.Ho, g.Wo, g.Cout, g.dy_pitch = Ho, Wo, cout, rup(cout, 8)
    g.in_stride, g.ntaps = stride, len(taps)
    for i, (a, b) in enumerate(taps):
        g.dy_t[i], g.dx_t[i] = a, b
    g.dw_cin, g.accumulate = cin, 1
    nbytes = int(nv.lib().hrp_wgrad_workspace_bytes(C.byref(g)))
    ws = torch.zeros(nbytes // 4 + 4, device=DEV)
    g.workspace, g.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    if TIMELINE:
        ws = torch.zeros(nbytes // 4 + (1 << 18) + 4, device=DEV)
        g.workspace, g.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    us = timeit(lambda: nv.call("hrp_conv2d_bwd_weight", C.byref(g), None))
    if TIMELINE:
        tl = ws.view(torch.int64)[-(1 << 17):].cpu().view(-1, 16)
        tl = tl[tl[:, 0] > 0].double()
        t0 = tl[:, 0].min()
        names = ["entry", "setup", "issued", "t0 ready", "t0 done", "t1 ready", "t1 done", "t2 ready", "t2 done",
                 "t3 ready", "t3 done", "loop end", "round1", "stored"]
        print(f"      timeline over {tl.shape[0]} workgroups (us after the first entry; mean / max):")
        print("      " + "  ".join(f"{n} {((tl[:, i] - t0).mean() / 100):.2f}/{((tl[:, i] - t0).max() / 100):.2f}"
                                  for i, n in enumerate(names) if tl[:, i].max() > 0))
    print(f"wgrad N={N} {cin:4d}->{cout:4d} k{k} s{stride} @{H:3d}x{W:<3d}         : {us:8.1f} us  "
          f"{fl / us / 1e6:7.1f} TFLOP/s  {by / us / 1e3:7.1f} GB/s")


def ew_case(N, H, W, Cc, dtype):
    esz = 2 if dtype == torch.bfloat16 else 4
    a = torch.randn(N * H * W * Cc, device=DEV).to(dtype)
    b = torch.randn(N * H * W * Cc, device=DEV).to(dtype)
    out = torch.zeros_like(a)
    st = torch.rand(16 * Cc, dtype=torch.float64, device=DEV) + 1.0      # statistic slots are fp64
    gam, bet = torch.ones(Cc, device=DEV), torch.zeros(Cc, device=DEV)
    d = nv.EwDesc()
    d.nin, d.out, d.out_pitch, d.dtype = 2, out.data_ptr(), Cc, nv.HRP_BF16 if esz == 2 else nv.HRP_F32
    d.N, d.H, d.W, d.C, d.relu = N, H, W, Cc, 1
    e = d.inp[0]
    e.ptr, e.pitch, e.up, e.mode = a.data_ptr(), Cc, 1, nv.EW_BN_TRAIN
    e.a, e.b, e.stats, e.count, e.eps = gam.data_ptr(), bet.data_ptr(), st.data_ptr(), float(N * H * W), 1e-5
    e = d.inp[1]
    e.ptr, e.pitch, e.up, e.mode = b.data_ptr(), Cc, 1, nv.EW_IDENTITY
    us = timeit(lambda: nv.call("hrp_ew_fwd", C.byref(d), None))
    by = 3 * a.numel() * esz
    print(f"ew_fwd (bn+res+relu) N={N} C={Cc:4d} @{H:3d}x{W:<3d}: {us:8.1f} us  {by / us / 1e3:7.1f} GB/s")
    bd = nv.EwBwdDesc()
    din = torch.zeros_like(a)
    sums = torch.zeros(16 * Cc, dtype=torch.float64, device=DEV)
    bd.dout, bd.out, bd.dout_pitch, bd.out_pitch = b.data_ptr(), out.data_ptr(), Cc, Cc
    for f, _ in nv.EwInput._fields_:
        setattr(bd.inp, f, getattr(d.inp[0], f))
    bd.din, bd.din_pitch, bd.sums, bd.dtype = din.data_ptr(), Cc, sums.data_ptr(), d.dtype
    bd.N, bd.H, bd.W, bd.C, bd.relu, bd.accumulate = N, H, W, Cc, 1, 0
    us = timeit(lambda: nv.call("hrp_ew_bwd_reduce", C.byref(bd), None))
    print(f"ew_bwd_reduce        N={N} C={Cc:4d} @{H:3d}x{W:<3d}: {us:8.1f} us  {by / us / 1e3:7.1f} GB/s")
    us = timeit(lambda: nv.call("hrp_ew_bwd_apply", C.byref(bd), None))
    print(f"ew_bwd_apply         N={N} C={Cc:4d} @{H:3d}x{W:<3d}: {us:8.1f} us  {4 * a.numel() * esz / us / 1e3:7.1f} GB/s")


def interleave_case(N, hw, c, dtype):
    """Cold-start effects: the same conv launched back to back vs alternating with an element-wise kernel and
    with a second conv of another template instance (different code, different weights)."""
    esz = 2
    x = torch.randn(N * hw * hw * c, device=DEV).to(dtype)
    y = torch.zeros_like(x)

    def mk(cin, cout, k):
        w = torch.randn(cout, cin, k, k, device=DEV) / (cin * k * k) ** 0.5
        wp, _ = pack(w, dtype)
        d = nv.ConvDesc()
        d.x, d.w, d.y = x.data_ptr(), wp.data_ptr(), y.data_ptr()
        d.dtype = nv.HRP_BF16
        d.N, d.H, d.W, d.Cin, d.x_pitch = N, hw, hw, cin, c
        d.Ho, d.Wo, d.Cout = hw, hw, cout
        d.y_H, d.y_W, d.y_pitch, d.res_pitch = hw, hw, c, c
        d.out_stride, d.in_stride = 1, 1
        taps = TAPS3 if k == 3 else [(0, 0)]
        d.ntaps = d.w_ntaps = len(taps)
        for i, (a, b) in enumerate(taps):
            d.dy[i], d.dx[i], d.wtap[i] = a, b, i
        d.w_cout_pad = rup(cout, 32)
        return d, wp

    d1, k1 = mk(c, c, 3)
    ds = [mk(c, c, 3) for _ in range(8)]          # same kernel, 8 different weight sets
    d2, k2 = mk(c, c, 1)                          # another template instance
    e = nv.EwDesc()
    out = torch.zeros_like(x)
    e.nin, e.out, e.out_pitch, e.dtype = 1, out.data_ptr(), c, nv.HRP_BF16
    e.N, e.H, e.W, e.C, e.relu = N, hw, hw, c, 1
    e.inp[0].ptr, e.inp[0].pitch, e.inp[0].up, e.inp[0].mode = y.data_ptr(), c, 1, nv.EW_IDENTITY
    t_conv = timeit(lambda: nv.call("hrp_conv2d_fwd", C.byref(d1), None))
    t_ew = timeit(lambda: nv.call("hrp_ew_fwd", C.byref(e), None))
    t_c1 = timeit(lambda: nv.call("hrp_conv2d_fwd", C.byref(d2), None))
    it = [0]

    def rot():
        nv.call("hrp_conv2d_fwd", C.byref(ds[it[0] % 8][0]), None)
        it[0] += 1
    t_rot = timeit(rot)
    t_alt = timeit(lambda: (nv.call("hrp_conv2d_fwd", C.byref(d1), None), nv.call("hrp_ew_fwd", C.byref(e), None)))
    t_alt2 = timeit(lambda: (nv.call("hrp_conv2d_fwd", C.byref(d1), None), nv.call("hrp_conv2d_fwd", C.byref(d2), None)))
    print(f"interleave C={c} @{hw}: conv3x3 {t_conv:.1f}  ew {t_ew:.1f}  conv1x1 {t_c1:.1f} | 8 weight sets {t_rot:.1f} | "
          f"conv3x3+ew {t_alt:.1f} (sum {t_conv + t_ew:.1f}) | conv3x3+conv1x1 {t_alt2:.1f} (sum {t_conv + t_c1:.1f})")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="?", default="all")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--dtype", default="bf16")
    a = ap.parse_args()
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    B = a.batch
    if a.what in ("conv", "wgrad", "all"):
        for (cin, cout, k, s, hw) in [(32, 32, 3, 1, 64), (64, 64, 3, 1, 32), (128, 128, 3, 1, 16), (256, 256, 3, 1, 8),
                                      (64, 64, 3, 1, 64), (64, 256, 1, 1, 64), (256, 64, 1, 1, 64), (32, 64, 3, 2, 64),
                                      (32, 448, 1, 1, 64), (1024, 2048, 1, 1, 8)]:
            for stats in ((False, True) if cin == 32 and cout == 32 else (True,)):
                conv_case(B, hw, hw, cin, cout, k, s, dt, stats)
    if a.what == "fc":     # the regression heads' fully connected layers on the conv path (fp32, 64 rows)
        for (cin, cout) in [(1024, 1024), (2056, 1024), (1024, 8)]:
            conv_case(B, 1, 1, cin, cout, 1, 1, torch.float32, False)
    if a.what == "interleave":
        for (c, hw) in [(32, 64), (64, 32), (128, 16), (256, 8)]:
            interleave_case(B, hw, c, dt)
    if a.what in ("ew", "all"):
        for (c, hw) in [(32, 64), (64, 32), (128, 16), (256, 8), (256, 64)]:
            ew_case(B, hw, hw, c, dt)
