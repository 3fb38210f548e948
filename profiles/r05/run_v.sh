mkdir -p gpurun_out/r05v
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_engine_gpu.py tests/test_fullwidth_gpu.py tests/test_step_cache_gpu.py -m gpu -x -q 2>&1 | tail -2
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | cut -c1-200
