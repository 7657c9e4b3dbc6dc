#!/usr/bin/env python3
"""Micro-benchmark of the skinny linear kernels (hrp_linear_*) on the regression heads' shapes (run on the GPU box)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_kernels as bk  # noqa: E402
from hrpe_amd import _native as nv  # noqa: E402

DEV = bk.DEV
for (M, K, N) in [(64, 2056, 1024), (64, 1024, 1024), (64, 1024, 8)]:
    x = torch.randn(M, K, device=DEV)
    w = torch.randn(N, K, device=DEV) / K ** 0.5
    b = torch.randn(N, device=DEV)
    y = torch.zeros(M, N, device=DEV)
    dx = torch.zeros(M, K, device=DEV)
    dw = torch.zeros(N, K, device=DEV)
    db = torch.zeros(N, device=DEV)
    wsb = int(nv.lib().hrp_linear_workspace_bytes(M, K, N))
    ws = torch.zeros(wsb // 4 + 4, device=DEV)
    t1a = bk.timeit(lambda: nv.call("hrp_linear_fwd", x.data_ptr(), K, w.data_ptr(), b.data_ptr(), None, 0, y.data_ptr(), N, M, K, N, None, 0, None))
    t2a = bk.timeit(lambda: nv.call("hrp_linear_bwd_data", y.data_ptr(), N, w.data_ptr(), dx.data_ptr(), K, M, K, N, 0, None, 0, None))
    t1 = bk.timeit(lambda: nv.call("hrp_linear_fwd", x.data_ptr(), K, w.data_ptr(), b.data_ptr(), None, 0, y.data_ptr(), N, M, K, N, ws.data_ptr(), wsb, None))
    t2 = bk.timeit(lambda: nv.call("hrp_linear_bwd_data", y.data_ptr(), N, w.data_ptr(), dx.data_ptr(), K, M, K, N, 0, ws.data_ptr(), wsb, None))
    print(f"   (fp32 atomics instead of the ordered reduction: fwd {t1a:6.1f} us  bwd_data {t2a:6.1f} us)")
    t3 = bk.timeit(lambda: nv.call("hrp_linear_bwd_weight", x.data_ptr(), K, y.data_ptr(), N, dw.data_ptr(), db.data_ptr(), M, K, N, 1, None))
    wb = N * K * 4
    print(f"linear M={M} K={K} N={N}: fwd {t1:6.1f} us ({wb / t1 / 1e3:6.0f} GB/s)  bwd_data {t2:6.1f} us  bwd_weight {t3:6.1f} us "
          f"({2 * wb / t3 / 1e3:6.0f} GB/s)")
