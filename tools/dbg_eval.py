"""Debug aid: first conv / activation output that differs between the batched and the one-by-one eval plan."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import hrpe_amd  # noqa
from hrpe_amd import plan as P
from synth import synth_inputs
from test_gpu_model import DEV, build_full

m = build_full().eval()
x_reg, x_root, kv, K = [t.to(DEV) for t in synth_inputs(2)]
REC = []
_conv, _act = P.PlanBuilder.conv, P.PlanBuilder.act


def conv(self, x, weight, *a, **k):
    y = _conv(self, x, weight, *a, **k)
    REC.append(("conv", tuple(weight.shape), k.get("stride", 1), y))
    return y


def act(self, terms, relu):
    y = _act(self, terms, relu)
    REC.append(("act", len(terms), [t.up for t in terms], y))
    return y


P.PlanBuilder.conv, P.PlanBuilder.act = conv, act
snap = {}
for name, batching in (("batched", True), ("single", False)):
    P.BATCHING = batching
    m.invalidate_plans()
    del REC[:]
    with torch.no_grad():
        m(x_reg, x_root, kv, K)
    torch.cuda.synchronize()
    snap[name] = [(r[0], r[1], r[2], r[3].buf.clone(), (r[3].N, r[3].H, r[3].W, r[3].C)) for r in REC]
n = 0
for a, b in zip(snap["batched"], snap["single"]):
    if not torch.equal(a[3], b[3]):
        d = (a[3].float() - b[3].float()).abs()
        print("DIFF", a[0], a[1], a[2], a[4], "max abs", float(d.max()), "frac", float((d > 0).float().mean()))
        n += 1
        if n > 12:
            break
print("records", len(snap["batched"]), "diffs shown", n)
