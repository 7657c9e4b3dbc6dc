#!/usr/bin/env python3
"""development tool (GPU box): the deferred weight-gradient fold launches of the benchmark network's training plan - per launch
the problems' (slabs G, 32 x 32 pair blocks, tile elements), workgroups and slab bytes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from hrpe_amd import plan as P  # noqa: E402

if __name__ == "__main__":
    dev = torch.device("cuda:0")
    model = bench.build_model(0.5).to(dev).set_compute_dtype(torch.bfloat16).train()
    from hrpe_amd.lib.core.function import compute_k_values
    d = {k: torch.tensor(v).to(dev) for k, v in bench.synthetic_batch(64, 4242).items()}
    K = d["K"]
    kv = compute_k_values(K[:, 0, 0], K[:, 1, 1], d["bbox"])
    out = model(d["x_reg"], d["x_root"], kv, K)
    sum(o.float().sum() for o in out).backward()
    plan = next(iter(model._plans.values())).plan
    tot_b = tot_w = 0
    for e in plan.bwd_ops():
        op = e.op
        if isinstance(op, P.BatchLaunch) and op.fam == "wgrad_fold":
            rows = [(it.desc.G, it.desc.pairs, it.desc.nte, it.desc.nb) for it in op.launches()]
            wgs = sum(nte * 4 * pairs for _, pairs, nte, _ in rows)
            mb = sum(G * pairs * nte * 4096 for G, pairs, nte, _ in rows) / 1e6
            tot_b += mb
            tot_w += wgs
            print(f"{len(rows):3d} problems {wgs:6d} wgs {mb:8.1f} MB slabs  " + " ".join(f"G{G}x{pairs}p x{nte}" for G, pairs, nte, _ in rows))
    print("total", tot_w, "wgs", tot_b, "MB")
