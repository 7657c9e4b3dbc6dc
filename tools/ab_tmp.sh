f() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l)
        print(j['value'], j['ms_per_step'], j['config']['workload'][:60], j.get('max_px_err'))
"; }
python bench.py --forward-only --workload hrnet --no-extra --no-cpu-baseline 2>&1 | tail -1 | f
HRP_BENCH_REG_BACKBONE=resnet50 python bench.py --no-extra --no-cpu-baseline 2>&1 | tail -1 | f
python bench.py --dtype fp32 --no-extra --no-cpu-baseline --steps 5 2>&1 | tail -1 | f
