#!/bin/bash
# development helper (GPU box): the default bench of this tree against a second tree (_old/: `git archive <commit> | tar -x -C _old` +
# make in its csrc), alternating, on ONE box.   tools/ab_tree.sh [rounds]
for i in $(seq ${1:-3}); do
  for t in _old .; do
    (cd $t && python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null) | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$t', d['value'], d['ms_per_step'])"
  done
done
