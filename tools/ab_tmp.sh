python -m pytest tests/test_gpu_pwconv.py -x -q 2>&1 | tail -25
HRP_BENCH_SHAPES=300 python bench.py --no-extra 2>&1 >/dev/null | grep "taps1 s1/1" | head -30
