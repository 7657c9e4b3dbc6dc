#!/usr/bin/env python3
"""Development aid: capture a forward+backward of a small model into a HIP graph with lanes enabled."""
import faulthandler
import os
import sys

import torch

faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hrpe_amd  # noqa: E402,F401
from hrpe_amd.lib.models.backbones.HRnet import get_hrnet  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
if os.environ.get("FULL"):
    import bench
    from hrpe_amd.lib.core.function import compute_k_values, full_loss
    m = bench.build_model(0.0).to(dev).set_compute_dtype(torch.bfloat16).train()
    d = {k: torch.tensor(v).to(dev) for k, v in bench.synthetic_batch(B, 808).items()}
    K = d["K"]
    kv = compute_k_values(K[:, 0, 0], K[:, 1, 1], d["bbox"])

    def step():
        pred = m(d["x_reg"], d["x_root"], kv, K)
        sum(p.float().mean() for p in pred).backward()
else:
    m = get_hrnet(32, 7, 64, pretrain=False, generate_feat=True, generate_hm=True).to(dev).set_compute_dtype(torch.bfloat16).train()
    x = torch.rand(B, 3, 256, 256, device=dev)

    def step():
        heat, feat = m(x)
        (heat.float().mean() + feat.mean()).backward()


step()
torch.cuda.synchronize()
print("eager ok", flush=True)
side = torch.cuda.Stream(dev)
side.wait_stream(torch.cuda.current_stream(dev))
with torch.cuda.stream(side):
    step()
torch.cuda.current_stream(dev).wait_stream(side)
torch.cuda.synchronize()
print("side ok", flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    step()
print("captured", flush=True)
g.replay()
torch.cuda.synchronize()
print("replayed", flush=True)
