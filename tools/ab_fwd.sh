#!/bin/bash
# development helper (GPU box): forward-only bench under several environment settings.   tools/ab_fwd.sh "A=1" "B=2 C=3" ...
for cfg in "$@"; do
  env $cfg python bench.py --forward-only --steps 30 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$cfg', d['value'], d['ms_per_step'], {k:(v['launches'],v['ms']) for k,v in list(d['kernels'].items())[:4]})"
done
