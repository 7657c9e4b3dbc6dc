#!/bin/bash
# development helper (GPU box): rocprofv3 kernel statistics of the forward-only bench -> gpurun_out/r04_fwd_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_fwd
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_fwd -- python3 $R/bench.py --forward-only --steps 20 --warmup 3 --no-cpu-baseline --no-extra > $R/gpurun_out/r04_fwd_prof_bench.json 2> $R/gpurun_out/r04_fwd_prof_bench.err
f=$(find /tmp/prof_fwd -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/gpurun_out/r04_fwd_kernel_stats.csv
head -30 $R/gpurun_out/r04_fwd_kernel_stats.csv | cut -c1-200
