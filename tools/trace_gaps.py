#!/usr/bin/env python3
"""development helper: from a rocprofv3 --kernel-trace CSV of bench.py, the busy / idle structure of the last replayed step:
wall time, union of kernel intervals, time with >= 2 kernels in flight, the largest idle gaps and what surrounds them."""
import csv
import sys

rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
rows.sort()
# one step = between two consecutive launches of the Adam kernel
marks = [i for i, r in enumerate(rows) if "opt_adam_kernel" in r[2]]
a, b = marks[-2] + 1, marks[-1] + 1
step = rows[a:b]
t0, t1 = step[0][0], max(r[1] for r in step)
ev = []
for s, e, n in step:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
busy = over2 = 0
depth, last = 0, t0
for t, d in ev:
    if depth >= 1: busy += t - last
    if depth >= 2: over2 += t - last
    depth += d; last = t
print(f"kernels {len(step)}  wall {(t1 - t0) / 1e6:.3f} ms  busy(union) {busy / 1e6:.3f} ms  >=2 in flight {over2 / 1e6:.3f} ms  "
      f"sum of durations {sum(e - s for s, e, _ in step) / 1e6:.3f} ms")
# idle gaps
gaps = []
end = step[0][1]
prev = step[0][2]
for s, e, n in step[1:]:
    if s > end:
        gaps.append((s - end, prev, n, (end - t0) / 1e6))
    if e > end:
        end, prev = e, n
gaps.sort(reverse=True)
print(f"idle total {sum(g[0] for g in gaps) / 1e6:.3f} ms in {len(gaps)} gaps; > 5 us: {sum(g[0] for g in gaps if g[0] > 5000) / 1e6:.3f} ms")
for g in gaps[:25]:
    print(f"  {g[0] / 1e3:7.1f} us at {g[3]:7.3f} ms  after {g[1]:45s} before {g[2]}")
