timeout 900 python -m pytest tests/test_step_cache_gpu.py tests/test_engine_gpu.py tests/test_fullwidth_gpu.py -m gpu -x -q 2>&1 | tail -2
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | cut -c1-260
