#!/usr/bin/env python3
"""development tool (GPU box): is the fp32 inference forward bit-reproducible?  Runs the full network, the DepthNet and a bare
HRNet several times (fresh model each time and repeated calls) and reports which outputs differ."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import test_gpu_model as M  # noqa: E402
from synth import synth_inputs, synth_state_dict  # noqa: E402

DEV = "cuda:0"
x_reg, x_root, kv, K = [t.to(DEV) for t in synth_inputs(2)]


def runs(make, call, n=6):
    outs = []
    for i in range(n):
        m = make() if i % 2 == 0 else m      # noqa: F821  (a fresh model every second call)
        with torch.no_grad():
            o = call(m)
        outs.append([t.detach().clone() for t in (o if isinstance(o, (tuple, list)) else [o])])
    return outs


def report(name, outs):
    nd = [sum(not torch.equal(o[k], outs[0][k]) for o in outs[1:]) for k in range(len(outs[0]))]
    mx = [max(float((o[k] - outs[0][k]).abs().max()) for o in outs[1:]) for k in range(len(outs[0]))]
    print(f"{name}: differing runs per output {nd}, max abs diff {['%.2e' % v for v in mx]}")


if __name__ == "__main__":
    from hrpe_amd.lib.models.backbones.HRnet import get_hrnet
    from hrpe_amd.lib.models.depth_net import get_rootnet

    def mk_hr():
        m = get_hrnet(32, 7, 64, pretrain=False, generate_feat=True, generate_hm=True)
        m.load_state_dict(synth_state_dict(m.state_dict()))
        return m.to(DEV).eval()
    report("HRNet (heat-map, feature)", runs(mk_hr, lambda m: m(x_reg)))

    def mk_root():
        m = get_rootnet("hrnet32")
        m.load_state_dict(synth_state_dict(m.state_dict()))
        return m.to(DEV).eval()
    report("DepthNet", runs(mk_root, lambda m: m(x_root, kv)))
    report("full network 8-tuple", runs(lambda: M.build_full().eval(), lambda m: m(x_reg, x_root, kv, K)))
