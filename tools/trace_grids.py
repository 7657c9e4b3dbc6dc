#!/usr/bin/env python3
"""Workgroups per launch from a rocprofv3 --kernel-trace CSV (which launches do not fill the 256 CUs?).

    python tools/trace_grids.py <kernel_trace.csv> [steps]

Per kernel: launches per step, the share of them below 256 workgroups, min / median / max workgroups, and the kernel time spent in
launches below 256 workgroups."""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    name = name.replace("hrp::", "")
    return re.sub(r"\(.*$", "", name)[:70]


def main():
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    rows = defaultdict(list)
    with open(sys.argv[1], newline="") as fh:
        for r in csv.DictReader(fh):
            wg = max(int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 256)) or 256), 1)
            grid = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)
            rows[short(r["Kernel_Name"])].append((grid // wg, int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    once = [k for k in rows if "softargmax_fwd_kernel" in k]      # one launch per step of the full network
    if once and len(sys.argv) <= 2:
        steps = float(len(rows[once[0]]))
    print(f"steps in the trace: {steps:.0f}")
    tot_small = tot = 0.0
    print(f"{'kernel':70s} {'launches/step':>13s} {'< 256 wgs':>10s} {'min':>6s} {'median':>7s} {'max':>7s} {'ms/step':>8s} {'ms/step < 256':>14s}")
    for k, v in sorted(rows.items(), key=lambda kv: -sum(d for _, d in kv[1])):
        if not (k.startswith(("conv_", "wgrad_", "ew_", "block_", "linear")) or "conv" in k):
            continue
        w = sorted(x for x, _ in v)
        small = [d for x, d in v if x < 256]
        t, ts = sum(d for _, d in v) / 1e6 / steps, sum(small) / 1e6 / steps
        tot += t
        tot_small += ts
        print(f"{k:70s} {len(v) / steps:13.1f} {len(small) / len(v):10.2f} {w[0]:6d} {w[len(w) // 2]:7d} {w[-1]:7d} {t:8.3f} {ts:14.3f}")
    print(f"kernel time in launches below 256 workgroups: {tot_small:.2f} of {tot:.2f} ms per step")


if __name__ == "__main__":
    main()
