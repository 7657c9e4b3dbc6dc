#!/usr/bin/env python3
"""Development tool: the element-wise kernels against torch's plain streaming kernels on the four branch tensors
(how far is hrp_ew_* from what a copy / add reaches at the same size?).  Run on the GPU box."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_kernels as bk  # noqa: E402

DEV = bk.DEV
for c, hw in [(32, 64), (64, 32), (128, 16), (256, 8)]:
    n = 64 * hw * hw * c
    a = torch.randn(n, device=DEV).bfloat16()
    b = torch.randn(n, device=DEV).bfloat16()
    o = torch.zeros_like(a)
    us = bk.timeit(lambda: o.copy_(a))
    print(f"C={c} @{hw}: torch copy {us:6.1f} us {2 * n * 2 / us / 1e3:7.1f} GB/s", end="  ")
    us = bk.timeit(lambda: torch.add(a, b, out=o))
    print(f"torch add {us:6.1f} us {3 * n * 2 / us / 1e3:7.1f} GB/s")
    bk.ew_case(64, hw, hw, c, torch.bfloat16)
