#!/usr/bin/env python3
"""Per-kernel means of SQ counters from a rocprofv3 --pmc run (several counters in one pass).

    python tools/pmc_sq_summary.py <dir with *counter_collection.csv> <out.json> [substring filter ...]

Ratios are taken against SQ_WAVE_CYCLES (quad-cycles summed over waves); SQ_VALU_MFMA_BUSY_CYCLES counts cycles, so
mfma_busy_frac = MFMA_BUSY / (4 * WAVE_CYCLES) is the share of wave time with the matrix pipe busy
(MI355X_MICROARCH.md, "rocprofv3 PMC slots").
"""
import collections
import csv
import glob
import json
import os
import sys


def main():
    d, out = sys.argv[1], sys.argv[2]
    filt = sys.argv[3:] or ["conv_tile", "conv_wgrad", "ew_"]
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f, newline="")):
        n = r["Kernel_Name"]
        if not any(k in n for k in filt):
            continue
        key = n.split("(")[0].replace("void hrp::", "") + " grid" + r["Grid_Size"]
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {}
    for k, v in agg.items():
        m = {c: sum(x) / len(x) for c, x in v.items()}
        wc = m.get("SQ_WAVE_CYCLES", 0.0) or 1.0
        e = {"dispatches": len(next(iter(v.values()))), "wave_quad_cycles": wc,
             "insts_valu": m.get("SQ_INSTS_VALU"), "insts_lds": m.get("SQ_INSTS_LDS"),
             "active_frac": m.get("SQ_ACTIVE_INST_ANY", 0) / wc, "wait_any_frac": m.get("SQ_WAIT_ANY", 0) / wc,
             "wait_inst_frac": m.get("SQ_WAIT_INST_ANY", 0) / wc, "wait_lds_frac": m.get("SQ_WAIT_INST_LDS", 0) / wc,
             "mfma_busy_frac": m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (4 * wc)}
        res[k] = e
        print(f"{k[:72]:72s} mfma {e['mfma_busy_frac']:.2f} active {e['active_frac']:.2f} wait {e['wait_any_frac']:.2f} valu {e['insts_valu']:.3g}")
    json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
