#!/bin/bash
# development aid: bench.py under a list of environment settings (one per line on stdin: "VAR=val VAR2=val"), prints
# ms_per_step and the kernel-time of the element-wise / conv / wgrad families
while read -r line; do
  out=$(env $line timeout 200 python bench.py --steps 12 --warmup 4 --no-cpu-baseline 2>/dev/null | tail -1)
  python - "$line" <<PY "$out"
import json, sys
try:
    d = json.loads(sys.argv[2]); k = d["kernels"]
    print(f"{sys.argv[1]:60s} ms {d['ms_per_step']:.2f}  conv {k['hrp_conv2d_fwd']['ms']:.2f} wgrad {k['hrp_conv2d_bwd_weight']['ms']:.2f} "
          f"apply {k['hrp_ew_bwd_apply']['ms']:.2f} red {k['hrp_ew_bwd_reduce']['ms']:.2f} fwd {k['hrp_ew_fwd']['ms']:.2f}")
except Exception as e:
    print(sys.argv[1], "FAILED", repr(e)[:100])
PY
done
