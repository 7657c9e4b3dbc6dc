#!/bin/bash
# development helper (GPU box): leave-one-out timing of the replayed step - bench.py with C-ABI entry points turned into no-ops
# (HRP_SKIP, _native.call): what a family really costs on the step's critical path.   tools/ab_skip.sh "" "hrp_a,hrp_b" ...
for cfg in "$@"; do
  HRP_SKIP="$cfg" python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('skip[$cfg]', d['value'], d['ms_per_step'])"
done
