mkdir -p gpurun_out/r06c
timeout 1500 python -m pytest tests/test_timestep_gpu.py tests/test_engine_gpu.py tests/test_step_cache_gpu.py tests/test_fullwidth_gpu.py -m gpu -q 2>&1 | tail -15
python profiles/shape_probe.py 128 bf16 2 cfg > gpurun_out/r06c/shape_b2_cfg.txt 2>&1; head -3 gpurun_out/r06c/shape_b2_cfg.txt
python profiles/shape_probe.py 64 bf16 2 cfg > gpurun_out/r06c/shape_b2_cfg_64.txt 2>&1; head -3 gpurun_out/r06c/shape_b2_cfg_64.txt
python profiles/shape_probe.py 64 bf16 2 > gpurun_out/r06c/shape_b2_64.txt 2>&1; head -3 gpurun_out/r06c/shape_b2_64.txt
