#!/usr/bin/env python3
"""development tool: from a rocprofv3 --kernel-trace CSV of bench.py, the wall time of the LAST replayed steps during which no
"big" kernel (>= BIG workgroups) is in flight - the serial head / optimizer sections where the chip idles behind tiny launches.

    python tools/trace_small.py <dir with *kernel_trace.csv> [steps=10] [big=128]
"""
import csv
import glob
import os
import sys
from collections import Counter


def main():
    d = sys.argv[1]
    nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    big = int(sys.argv[3]) if len(sys.argv) > 3 else 128
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = []
    with open(f, newline="") as fh:
        for r in csv.DictReader(fh):
            wgs = (int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)) * max(int(r.get("Grid_Size_Y", 1)) // max(int(r.get("Workgroup_Size_Y", 1)), 1), 1)
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], wgs))
    rows.sort()
    # steps: delimited by the optimizer's Adam kernel (one per step)
    marks = [e for s, e, n, w in rows if "opt_adam" in n or "adam" in n.lower()]
    if len(marks) < nsteps + 1:
        print("too few steps in the trace", len(marks))
        return
    t0, t1 = marks[-nsteps - 1], marks[-1]
    sel = [r for r in rows if r[0] >= t0 and r[1] <= t1]
    ev = []
    for s, e, n, w in sel:
        ev.append((s, 1, w >= big, n))
        ev.append((e, -1, w >= big, n))
    ev.sort(key=lambda x: (x[0], x[1]))
    nb = na = 0
    last = t0
    t_none = t_small = t_big = 0
    small_names = Counter()
    for t, dlt, isbig, n in ev:
        dt = t - last
        if nb > 0:
            t_big += dt
        elif na > 0:
            t_small += dt
        else:
            t_none += dt
        last = t
        na += dlt
        if isbig:
            nb += dlt
    for s, e, n, w in sel:
        if w < big:
            small_names[n.split("(")[0][-60:]] += e - s
    span = (t1 - t0) / nsteps
    print(f"{nsteps} steps of {span / 1e6:.3f} ms: a kernel of >= {big} workgroups in flight {t_big / nsteps / 1e6:.3f} ms, only smaller ones "
          f"{t_small / nsteps / 1e6:.3f} ms, nothing {t_none / nsteps / 1e6:.3f} ms")
    for n, v in small_names.most_common(14):
        print(f"   {v / nsteps / 1e3:8.1f} us/step  {n}")


if __name__ == "__main__":
    main()
