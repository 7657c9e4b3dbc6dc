#!/usr/bin/env python3
"""One eager forward + loss + backward of the benchmark network (lanes folded onto one stream): the workload
of the rocprofv3 --pmc passes (profiles/r01_pmc_*), small enough for per-dispatch counter collection."""
import os
import sys

os.environ.setdefault("HRP_SERIAL_LANES", "1")
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from hrpe_amd.lib.core.function import compute_k_values  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
m = bench.build_model(0.5).to(dev).set_compute_dtype(torch.bfloat16).train()
d = {k: torch.tensor(v).to(dev) for k, v in bench.synthetic_batch(B, 808).items()}
K = d["K"]
kv = compute_k_values(K[:, 0, 0], K[:, 1, 1], d["bbox"])
pred = m(d["x_reg"], d["x_root"], kv, K)
sum(p.float().mean() for p in pred).backward()
torch.cuda.synchronize()
print("one step done")
