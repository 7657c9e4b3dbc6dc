#!/bin/bash
# Run on the GPU box (gpurun): rocprofv3 kernel statistics of the benchmark command and the two PMC passes (separate runs,
# no trace domains next to --pmc) the roofline.traffic figure comes from.  Writes under gpurun_out/ (scratch); the
# summaries judged are copied into profiles/ by hand.   ROUND=r05 [SUFFIX=_x HRP_...=..] bash tools/collect_profiles.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
rm -rf /tmp/prof_stats /tmp/prof_fetch /tmp/prof_write
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python3 $R/bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-extra > $O/${ROUND:-r05}${SUFFIX}_prof_bench.json 2> $O/${ROUND:-r05}${SUFFIX}_prof_bench.err
f=$(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${ROUND:-r05}${SUFFIX}_bench_kernel_stats.csv
f=$(find /tmp/prof_stats -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python3 $R/tools/trace_grids.py $f > $O/${ROUND:-r05}${SUFFIX}_launch_grids.txt
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof_fetch -- python3 $R/tools/one_step.py 64 > $O/${ROUND:-r05}${SUFFIX}_pmc_fetch.log 2>&1
python3 $R/tools/pmc_summary.py /tmp/prof_fetch FETCH_SIZE $O/${ROUND:-r05}${SUFFIX}_pmc_step_FETCH_SIZE.json | tail -5
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/prof_write -- python3 $R/tools/one_step.py 64 > $O/${ROUND:-r05}${SUFFIX}_pmc_write.log 2>&1
python3 $R/tools/pmc_summary.py /tmp/prof_write WRITE_SIZE $O/${ROUND:-r05}${SUFFIX}_pmc_step_WRITE_SIZE.json | tail -5
cd $R && python3 $R/tools/traffic_from_pmc.py $O/${ROUND:-r05}${SUFFIX}_pmc_step_FETCH_SIZE.json $O/${ROUND:-r05}${SUFFIX}_pmc_step_WRITE_SIZE.json $O/${ROUND:-r05}${SUFFIX}_traffic.json
head -25 $O/${ROUND:-r05}${SUFFIX}_bench_kernel_stats.csv | cut -c1-180
# SQ occupancy / stall counters of the same step (one pass, 8 SQ slots + GRBM): what the dominant kernels wait for
rm -rf /tmp/prof_sq
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d /tmp/prof_sq -- python3 $R/tools/one_step.py 64 > $O/${ROUND:-r05}${SUFFIX}_pmc_sq.log 2>&1
python3 $R/tools/pmc_sq_summary.py /tmp/prof_sq $O/${ROUND:-r05}${SUFFIX}_pmc_sq_step.json conv_batch wgrad_batch ew_ conv_tile conv_pw | sort -k2 | tail -40
