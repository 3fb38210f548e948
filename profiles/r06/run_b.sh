# round 6, call 2: timestep exactness + shared CFG prefix: tests, then same-box A/B
mkdir -p gpurun_out/r06b
timeout 1500 python -m pytest tests/test_timestep_gpu.py tests/test_engine_gpu.py tests/test_step_cache_gpu.py tests/test_fullwidth_gpu.py -m gpu -x -q 2>&1 | tail -15
python profiles/r06/share_ab.py 20 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06b/share_ab.txt
python profiles/shape_probe.py 128 bf16 2 > gpurun_out/r06b/shape_b2.txt 2>&1; head -3 gpurun_out/r06b/shape_b2.txt
