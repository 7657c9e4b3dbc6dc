#!/bin/bash
# development helper (GPU box): the metric's literal workload (one HRNet-W32 training step) under several environment settings
for cfg in "$@"; do
  env $cfg python bench.py --workload hrnet --steps 20 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$cfg', d['value'], d['ms_per_step'], {k:(v['launches'],v['ms']) for k,v in list(d['kernels'].items())[:3]})"
done
