#!/bin/bash
# Run on the GPU box (gpurun): rocprofv3 kernel statistics of the benchmark command and the two PMC passes (separate runs,
# no trace domains next to --pmc) the roofline.traffic figure comes from.  Writes under gpurun_out/ (scratch); the
# summaries judged are copied into profiles/ by hand.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
rm -rf /tmp/prof_stats /tmp/prof_fetch /tmp/prof_write
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python3 $R/bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-extra > $O/r04_prof_bench.json 2> $O/r04_prof_bench.err
f=$(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/r04_bench_kernel_stats.csv
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof_fetch -- python3 $R/tools/one_step.py 64 > $O/r04_pmc_fetch.log 2>&1
python3 $R/tools/pmc_summary.py /tmp/prof_fetch FETCH_SIZE $O/r04_pmc_step_FETCH_SIZE.json | tail -5
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/prof_write -- python3 $R/tools/one_step.py 64 > $O/r04_pmc_write.log 2>&1
python3 $R/tools/pmc_summary.py /tmp/prof_write WRITE_SIZE $O/r04_pmc_step_WRITE_SIZE.json | tail -5
python3 $R/tools/traffic_from_pmc.py $O/r04_pmc_step_FETCH_SIZE.json $O/r04_pmc_step_WRITE_SIZE.json $O/r04_traffic.json
head -25 $O/r04_bench_kernel_stats.csv | cut -c1-180
