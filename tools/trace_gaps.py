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
# the serial section between the trunks (everything from the first soft-argmax launch to the first backward element-wise launch
# behind it): what runs there, in order
if len(sys.argv) > 2 and sys.argv[2] == "heads":
    i0 = next(i for i, r in enumerate(step) if "softargmax_fwd" in r[2]) - 30
    i1 = next(i for i, r in enumerate(step) if "softargmax_bwd" in r[2]) + 30
    prev_end = step[i0][0]
    for s, e, n in step[i0:i1]:
        print(f"  {(s - t0) / 1e6:8.3f} ms  +{(s - prev_end) / 1e3:6.1f} us gap  {(e - s) / 1e3:7.1f} us  {n}")
        prev_end = max(prev_end, e)
    import collections
    c = collections.Counter(n for _, _, n in step)
    print({k: v for k, v in c.items() if "rocclr" in k or "at::" in k})
if len(sys.argv) > 2 and sys.argv[2] == "edges":
    for part in (step[:45], step[-40:]):
        prev_end = part[0][0]
        for s, e, n in part:
            print(f"  {(s - t0) / 1e6:8.3f} ms  +{(s - prev_end) / 1e3:6.1f} us gap  {(e - s) / 1e3:7.1f} us  {n}")
            prev_end = max(prev_end, e)
        print("  ...")
