#!/usr/bin/env python3
"""Development aid: the loss curves of tests/test_gpu_round4.py::test_bf16_training_follows_the_fp32_loss_curve with the round-6
fusions on and off (fused regressors, pooled fuse-sum gradients), to tell a training-dynamics change from run-to-run chaos."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import test_gpu_round4 as T  # noqa: E402
from hrpe_amd import plan as P  # noqa: E402
from hrpe_amd.lib.models import full_net as FN  # noqa: E402

n, nb, vis = 80, 4, 5
LR = float(os.environ.get("LR", "1e-4"))
for name, fused, pool in (("new", True, True), ("new again", True, True), ("old both", False, False)):
    FN.FUSED_REGRESSORS, P.POOL_FUSE_GRADS = fused, pool
    for dt in (torch.float32, torch.bfloat16):
        v = T._train_curve(dt, n, nbatches=nb, lr=LR)
        med = [[float(np.median(v[k::nb][w * vis:(w + 1) * vis])) for k in range(nb)] for w in range(n // (nb * vis))]
        fall = float(np.mean([med[-1][k] / med[0][k] for k in range(nb)]))
        print(f"{name:10s} {str(dt)[6:]:9s} fall {fall:.3f}  medians per window: " + " | ".join(" ".join(f"{x:7.1f}" for x in m) for m in med), flush=True)
