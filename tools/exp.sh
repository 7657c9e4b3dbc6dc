run() { name=$1; shift; env "$@" HRP_PLAN_STATS=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r02_x_$name.json 2> gpurun_out/r02_x_$name.err; python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r02_x_$name.json")); print("$name", d["value"], d["ms_per_step"], {k:(v["launches"],v["ms"]) for k,v in list(d["kernels"].items())[:5]})
except Exception as e: print("$name FAILED", e)
PY
}
run flat3 HRP_TRUNK_LANES=flat3
run flat3_np HRP_TRUNK_LANES=flat3 HRP_CONV_BATCH_PERSIST=0
