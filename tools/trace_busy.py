#!/usr/bin/env python3
"""Busy / concurrency summary of a rocprofv3 --kernel-trace CSV: for the last N ms of the trace, the fraction of
time with >= 1, 2, 4 kernels in flight and the per-family share of kernel-seconds.

    python tools/trace_busy.py <dir with *kernel_trace.csv> [window_ms] [skip_ms at the end]
"""
import csv
import glob
import os
import sys
from collections import Counter


def main():
    d = sys.argv[1]
    win = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 200e6
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = []
    with open(f, newline="") as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    skip = float(sys.argv[3]) * 1e6 if len(sys.argv) > 3 else 0.0
    t_end = max(r[1] for r in rows) - skip
    rows = [r for r in rows if r[0] >= t_end - win and r[1] <= t_end]
    t0 = rows[0][0]
    t_end = max(r[1] for r in rows)
    ev = []
    for s, e, _ in rows:
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    depth, last, hist = 0, t0, Counter()
    for t, dlt in ev:
        hist[depth] += t - last
        last = t
        depth += dlt
    span = t_end - t0
    print(f"window {span / 1e6:.1f} ms, {len(rows)} kernels")
    cum = 0
    for k in sorted(hist):
        print(f"  {k:2d} kernels in flight: {100 * hist[k] / span:5.1f}%")
    fam = Counter()
    for s, e, n in rows:
        key = next((k for k in ("conv_tile", "conv_wgrad", "wgrad_reduce", "ew_fwd", "ew_bwd_reduce", "ew_bwd_apply", "opt_", "pack_weights") if k in n), "other")
        fam[key] += e - s
    tot = sum(fam.values())
    print(f"kernel-seconds / span = {tot / span:.2f}")
    for k, v in fam.most_common():
        print(f"  {k:16s} {100 * v / tot:5.1f}%  ({v / 1e6:.1f} ms)")


if __name__ == "__main__":
    main()
