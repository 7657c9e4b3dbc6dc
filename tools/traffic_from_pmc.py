#!/usr/bin/env python3
"""profiles/r01_traffic.json from the two PMC summaries of tools/pmc_summary.py.

    python tools/traffic_from_pmc.py <FETCH_SIZE summary.json> <WRITE_SIZE summary.json> <out.json>

HBM bytes of a kernel family = (2 * FETCH_SIZE + WRITE_SIZE) KiB: both counters are reported in KiB, FETCH_SIZE is
doubled on gfx950 (128-byte requests counted as 64 bytes, MI355X_MICROARCH.md, HBM section).  Families are the kernel
names up to the template arguments; bench.py divides the conv family's bytes by its launches for `roofline.traffic`."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def fam(name):
    return name.split("<")[0].replace("hrp::", "").strip()


def main():
    f, w, out = [json.load(open(p)) if i < 2 else p for i, p in enumerate(sys.argv[1:4])]
    assert f["counter"] == "FETCH_SIZE" and w["counter"] == "WRITE_SIZE"
    fams = {}
    for src, key in ((f, "fetch_kb_raw"), (w, "write_kb")):
        for name, v in src["kernels"].items():
            e = fams.setdefault(fam(name), {"launches": 0, "fetch_kb_raw": 0.0, "write_kb": 0.0})
            e[key] += v["sum"]
            if key == "fetch_kb_raw":
                e["launches"] += v["dispatches"]
    for e in fams.values():
        e["hbm_bytes_per_step"] = (2.0 * e["fetch_kb_raw"] + e["write_kb"]) * 1024.0
        e["hbm_bytes_per_launch"] = e["hbm_bytes_per_step"] / max(e["launches"], 1)
    import hrpe_amd  # noqa: F401
    from hrpe_amd import _native as nv
    res = {
        # sha256[:16] of the library sources the counters were collected on: bench.py reports `roofline.traffic` only while the
        # library it measures carries the same hash
        "source_hash": nv.source_hash(),
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over tools/one_step.py: one eager "
                  "forward+loss+backward of the benchmark network, B=64, bf16, lanes folded onto one stream",
        "correction": "FETCH_SIZE and WRITE_SIZE are reported in KiB; FETCH_SIZE doubled (gfx950 counts 128-byte "
                      "requests as 64 bytes, MI355X_MICROARCH.md, HBM section); WRITE_SIZE as reported",
        "families": dict(sorted(fams.items(), key=lambda kv: -kv[1]["hbm_bytes_per_step"])),
        "step_hbm_bytes": sum(e["hbm_bytes_per_step"] for e in fams.values()),
        # the process also BUILDS the plan: torch's fill kernels zero every arena once (at::native::*), not part of a step
        "step_hbm_bytes_without_plan_build_fills": sum(e["hbm_bytes_per_step"] for k, e in fams.items() if not k.startswith("at::")),
    }
    json.dump(res, open(out, "w"), indent=1)
    for k, e in list(res["families"].items())[:8]:
        print(f"{k:28s} {e['launches']:5d} launches  {e['hbm_bytes_per_launch'] / 1e6:8.1f} MB / launch")
    print(f"step: {res['step_hbm_bytes'] / 1e9:.1f} GB ({res['step_hbm_bytes_without_plan_build_fills'] / 1e9:.1f} GB without the plan-build fills)")


if __name__ == "__main__":
    main()
