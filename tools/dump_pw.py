#!/usr/bin/env python3
"""development helper: the 1x1 stride-1 convolutions of the benchmark step and the descriptor options they use."""
import collections
import os
import sys

os.environ.setdefault("HRP_SERIAL_LANES", "1")
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from hrpe_amd import plan as P  # noqa: E402
from hrpe_amd.lib.core.function import compute_k_values  # noqa: E402

dev = torch.device("cuda:0")
m = bench.build_model(0.5).to(dev).set_compute_dtype(torch.bfloat16).train()
d = {k: torch.tensor(v).to(dev) for k, v in bench.synthetic_batch(64, 808).items()}
K = d["K"]
kv = compute_k_values(K[:, 0, 0], K[:, 1, 1], d["bbox"])
pred = m(d["x_reg"], d["x_root"], kv, K)
sum(p.float().mean() for p in pred).backward()
torch.cuda.synchronize()
cnt = collections.Counter()
for mod in m.modules():
    for r in getattr(mod, "_plans", {}).values():
        for name, lst in (("fwd", r.plan.fwd), ("bwd", r.plan.bwd)):
            for e in lst:
                op = getattr(e, "op", e)
                for it in (op.launches() if hasattr(op, "launches") else []):
                    if it.fam != "conv":
                        continue
                    c = it.desc
                    if c.ntaps != 1 or c.in_stride != 1 or c.out_stride != 1:
                        continue
                    flags = "".join(f for f, on in (("B", c.bias), ("A", c.scale), ("R", c.res), ("r", c.relu), ("S", c.stats),
                                                    ("b", c.bnb_x), ("m", c.bnb_mask), ("U", getattr(c, "up", 0))) if on)
                    dense = (c.x_pitch == c.Cin and c.y_pitch == c.Cout and (c.y_H, c.y_W) == (c.Ho, c.Wo))
                    cnt[(name, c.Cin, c.Cout, c.Ho, c.Wo, c.dtype, flags, dense, isinstance(op, P.BatchLaunch))] += 1
for k, v in sorted(cnt.items(), key=lambda kv: -kv[0][3] * kv[0][4] * (kv[0][1] + kv[0][2])):
    print(v, k)
