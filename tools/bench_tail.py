#!/usr/bin/env python3
"""Micro-benchmark of the Bottleneck-tail launches (hrp_conv_desc.tail_mode, csrc/conv_pw.h) at the shapes of the training step
(B = 64: 262 144 pixels; run on the GPU box): every mode alone, beside the launches it replaces - the plain pointwise conv with
statistics and the element-wise passes of the block end.  Prints microseconds, the bytes each launch must move, and TB/s."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hrpe_amd  # noqa: F401,E402
from hrpe_amd import _native as nv  # noqa: E402
import bench_kernels as bk  # noqa: E402

DEV = "cuda:0"


def main():
    for cin, cout in ((64, 256), (32, 128)):
        npix = 64 * 64 * 64
        g = torch.Generator(device="cpu").manual_seed(1)
        rb = lambda *s: torch.randn(*s, generator=g).to(DEV).to(torch.bfloat16)   # noqa: E731
        h, x2 = rb(npix, cin), rb(npix, cin)
        w = (torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5).to(DEV)
        wp, _ = bk.pack(w, torch.bfloat16)
        xs, dout = rb(npix, cout), rb(npix, cout)
        out, dy, side = torch.zeros_like(xs), torch.zeros_like(xs), torch.zeros_like(xs)
        gamma, beta = torch.ones(cout, device=DEV), torch.zeros(cout, device=DEV)
        stats = torch.zeros(8 * 2 * cout, dtype=torch.float64, device=DEV)
        bsums = torch.zeros(8 * 2 * cout, dtype=torch.float64, device=DEV)
        mask = torch.zeros(npix * cout // 8, dtype=torch.uint8, device=DEV)
        d = nv.ConvDesc()
        d.x, d.w, d.y, d.dtype = h.data_ptr(), wp.data_ptr(), out.data_ptr(), nv.HRP_BF16
        d.N, d.H, d.W, d.Cin, d.x_pitch = 64, 64, 64, cin, cin
        d.Ho, d.Wo, d.Cout, d.y_H, d.y_W, d.y_pitch, d.res_pitch = 64, 64, cout, 64, 64, cout, cout
        d.out_stride, d.in_stride, d.ntaps, d.w_ntaps, d.w_cout_pad = 1, 1, 1, 1, cout
        d.tail_gamma, d.tail_beta, d.tail_count, d.tail_eps, d.tail_mask = gamma.data_ptr(), beta.data_ptr(), float(npix), 1e-5, mask.data_ptr()

        def mk(mode, **kw):
            q = nv.ConvDesc()
            C.memmove(C.byref(q), C.byref(d), C.sizeof(nv.ConvDesc))
            q.tail_mode = mode
            for k, v in kw.items():
                setattr(q, k, v)
            assert mode == 0 or nv.lib().hrp_conv_pointwise(C.byref(q)) == 1, mode
            return q
        inb, outb = npix * cin * 2, npix * cout * 2
        cases = [("plain conv + stats", mk(0, stats=stats.data_ptr()), inb + outb),
                 ("mode 1 statistics", mk(1, stats=stats.data_ptr()), inb),
                 ("mode 2 fused forward", mk(2, res=xs.data_ptr(), tail_stats=stats.data_ptr()), inb + 2 * outb + outb // 16),
                 ("mode 5 projection forward", mk(5, tail_stats=stats.data_ptr(), tail_x2=x2.data_ptr(), tail_w2=wp.data_ptr(), tail_stats2=stats.data_ptr(),
                                                  tail_gamma2=gamma.data_ptr(), tail_beta2=beta.data_ptr()), 2 * inb + outb + outb // 16),
                 ("mode 3 backward reduce", mk(3, stats=bsums.data_ptr(), tail_stats=stats.data_ptr(), tail_g=dout.data_ptr()), inb + outb + outb // 16),
                 ("mode 4 backward apply", mk(4, y=dy.data_ptr(), tail_stats=stats.data_ptr(), tail_bsums=bsums.data_ptr(), tail_g=dout.data_ptr()),
                  inb + 2 * outb + outb // 16),
                 ("mode 4 + rider (acc)", mk(4, y=dy.data_ptr(), tail_stats=stats.data_ptr(), tail_bsums=bsums.data_ptr(), tail_g=dout.data_ptr(),
                                             tail_side=side.data_ptr(), tail_side_acc=1), inb + 4 * outb + outb // 16)]
        nv.call("hrp_conv2d_fwd", C.byref(cases[1][1]), None)      # real statistics for the later modes
        print(f"--- {cin} -> {cout} @ 64 x 64 x 64 images")
        for name, q, nbytes in cases:
            t = bk.timeit(lambda q=q: nv.call("hrp_conv2d_fwd", C.byref(q), None))
            print(f"{name:28s} {t:8.1f} us  {nbytes / 1e6:7.1f} MB  {nbytes / t / 1e6:5.2f} TB/s")


if __name__ == "__main__":
    main()
