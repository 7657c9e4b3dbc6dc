#!/bin/bash
# development aid: run one test N times in fresh processes under an environment setting, count failures
# usage: flake.sh N "<pytest -k expr>" [VAR=val ...]
n=$1; k=$2; shift 2
f=0
for i in $(seq 1 $n); do
  out=$(env "$@" timeout 120 python -m pytest ${HRP_FLAKE_FILE:-tests/test_gpu_kernels.py} -m gpu -q -x -k "$k" 2>&1 | grep -E "^E  .*Error|passed|failed" | cut -c1-220 | tr '\n' ' ')
  case "$out" in *failed*) f=$((f+1)); echo "  run $i: $out";; esac
done
echo "$* : $f / $n failed"
