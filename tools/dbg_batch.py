"""Debug aid: batched vs one-by-one plans on the reference's B = 8 training-step fixture - which gradients differ."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import hrpe_amd  # noqa
from hrpe_amd import plan as P
from hrpe_amd.lib.core.function import full_loss
from test_gpu_model import build_full, load
from test_gpu_parity import _train_step_inputs

g = load("golden_full_train_b8.npz")
m = build_full().train()
x_reg, x_root, kv, K, gt = _train_step_inputs(g, m, 8)


def step():
    m.zero_grad()
    pred = m(x_reg, x_root, kv, K)
    loss, _ = full_loss(pred, gt, K)
    loss.backward()
    torch.cuda.synchronize()
    return [p.detach().clone() for p in pred], {n: p.grad.clone() for n, p in m.named_parameters()}


REG = {}
_avg, _newlike, _cat = P.PlanBuilder.avgpool, P.PlanBuilder.new_like, P.PlanBuilder.cat_cols


def avgpool(self, x, out=None):
    y = _avg(self, x, out)
    REG.setdefault("avg", []).append((x, y))
    return y


def new_like(self, t):
    y = _newlike(self, t)
    REG.setdefault("newlike", []).append(y)
    return y


def cat_cols(self, parts, width=None):
    y = _cat(self, parts, width)
    REG.setdefault("cat", []).append(y)
    return y


P.PlanBuilder.avgpool, P.PlanBuilder.new_like, P.PlanBuilder.cat_cols = avgpool, new_like, cat_cols


def grad_norms():
    out = {}
    for i, (x, y) in enumerate(REG.get("avg", [])):
        out[f"avg{i}.in.grad"] = float(x.grad_buf().float().norm())
        out[f"avg{i}.out.grad"] = float(y.grad_buf().float().norm())
    for i, y in enumerate(REG.get("newlike", [])):
        out[f"newlike{i}.grad"] = float(y.grad_buf().float().norm())
    for i, y in enumerate(REG.get("cat", [])):
        out[f"cat{i}.grad"] = float(y.grad_buf().float().norm())
    cats = REG.get("cat", [])
    nl = REG.get("newlike", [])
    if len(cats) >= 8 and len(nl) >= 2:
        for lane, (x, cs) in enumerate(((nl[0], cats[0:4]), (nl[1], cats[4:8]))):
            g = x.grad_buf().view(x.N, x.pitch)[:, :2048].double()
            parts = [c.grad_buf().view(c.N, c.pitch)[:, :2048].double() for c in cs]
            A = torch.stack([p_.reshape(-1) for p_ in parts], 1)
            coef = torch.linalg.lstsq(A, g.reshape(-1, 1)).solution.reshape(-1)
            out[f"lane{lane}.coef"] = " ".join(f"{c:.3f}" for c in coef.tolist())
    return out


res = {}
norms = {}
seq = os.environ.get("DBG_SEQ", "batched,batched_same,single,batched2").split(",")
for name in seq:
    P.BATCHING = not name.startswith("single")
    if not name.endswith("_same"):
        m.invalidate_plans()
        REG.clear()
    res[name] = step()
    norms[name] = grad_norms()
rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-30))
import collections
for other in [n for n in seq if n != "single"]:
    print("==", other, "vs single: forward", [f"{rel(a, b):.1e}" for a, b in zip(res[other][0], res["single"][0])])
    grp = collections.OrderedDict()
    for n in res["single"][1]:
        if res["single"][1][n].norm() < 1e-10 or n.endswith(".bias") and ("downsamp" in n or "final_feat" in n):
            continue
        key = ".".join(n.split(".")[:3]) if "stage" in n else ".".join(n.split(".")[:2])
        e = rel(res[other][1][n], res["single"][1][n])
        cur = grp.get(key, (0.0, ""))
        if e > cur[0]:
            grp[key] = (e, n)
    for k, (e, n) in grp.items():
        if e > 3e-2:
            print(f"   {k:45s} {e:.2e}  {n}")

keys = list(norms["single"].keys())
for k in keys:
    print(f"{k:22s}", "  ".join(f"{n}={norms[n].get(k, float('nan'))}" for n in seq))
