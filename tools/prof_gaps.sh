#!/bin/bash
# development helper (GPU box): kernel trace of the default bench -> busy / idle structure of the last replayed step
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_tr
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tr -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-extra > /dev/null 2> /dev/null
f=$(find /tmp/prof_tr -name "*kernel_trace.csv" | head -1)
python3 $R/tools/trace_gaps.py $f $1 | head -${2:-40}
