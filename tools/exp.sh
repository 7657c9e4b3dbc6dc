# development aid: one gpurun call = tests + the bench lines of every workload (written under gpurun_out/)
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r02_tests_full.log 2>&1; grep -E "^E  |passed|failed|^FAILED" gpurun_out/r02_tests_full.log | cut -c1-300 | head -20 > gpurun_out/r02_tests.log
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r02_full.json 2> gpurun_out/bench_r02_full.err
timeout 300 python bench.py --steps 20 --warmup 5 --workload hrnet > gpurun_out/bench_r02_hrnet.json 2> gpurun_out/bench_r02_hrnet.err
timeout 300 python bench.py --steps 20 --warmup 5 --forward-only > gpurun_out/bench_r02_fwd.json 2> gpurun_out/bench_r02_fwd.err
timeout 300 python bench.py --steps 20 --warmup 5 --forward-only --workload hrnet > gpurun_out/bench_r02_fwd_hrnet.json 2> gpurun_out/bench_r02_fwd_hrnet.err
HRP_BENCH_DEVICE=0 HRP_DIST_BACKEND=gloo timeout 400 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_r02_gloo2.json 2> gpurun_out/bench_r02_gloo2.err
cat gpurun_out/r02_tests.log
for f in full hrnet fwd fwd_hrnet gloo2; do python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/bench_r02_$f.json").read().strip().splitlines()[-1])
    print("$f", d["value"], d["ms_per_step"], d.get("ms_per_step_without_optimizer"), d["n_gpus"], d.get("max_px_err"), d["roofline"]["kernel"], d["roofline"]["frac"], (d.get("cpu_baseline") or {}).get("value"))
except Exception as e: print("$f FAILED", repr(e)[:200])
PY
done
