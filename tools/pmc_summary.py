#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection.csv into per-kernel means.

    python tools/pmc_summary.py <dir with *counter_collection.csv> <counter name> <out.json>

Writes {kernel name: {"dispatches": n, "mean": mean counter value per dispatch, "sum": total}}.
"""
import csv
import glob
import json
import os
import sys


def main():
    d, counter, out = sys.argv[1], sys.argv[2], sys.argv[3]
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {d}")
    agg = {}
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                name = row["Kernel_Name"]
                short = name.split("(")[0].replace("void ", "")
                e = agg.setdefault(short, [0, 0.0])
                e[0] += 1
                e[1] += float(row["Counter_Value"])
    res = {k: {"dispatches": v[0], "mean": v[1] / v[0], "sum": v[1]} for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])}
    with open(out, "w") as fh:
        json.dump({"counter": counter, "kernels": res}, fh, indent=1)
    for k, v in list(res.items())[:12]:
        print(f"{v['dispatches']:6d} x {v['mean']:14.1f}  {k[:100]}")


if __name__ == "__main__":
    main()
