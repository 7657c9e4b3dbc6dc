for cfg in "HRP_ROWBW_FUSE=0" "HRP_ROWBW_FUSE=1" "HRP_ROWBW_FUSE=1 HRP_ROWBW_WGS=224"; do
  env $cfg python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$cfg', d['value'], d['ms_per_step'], {k:(v['launches'],v['ms']) for k,v in list(d['kernels'].items())[:4]})"
done
