#!/usr/bin/env python3
"""End-to-end key-point error in pixels of the bf16 path on the reference's eval fixture (development tool, GPU box)."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
print(json.dumps(bench.keypoint_px_error(torch.device("cuda:0"), [("fp32", torch.float32), ("bf16", torch.bfloat16)])))
