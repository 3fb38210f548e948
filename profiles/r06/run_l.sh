mkdir -p gpurun_out/r06l
timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_timestep_gpu.py tests/test_step_cache_gpu.py -m gpu -q 2>&1 | tail -3
( python profiles/r06/gemm_sweep_probe.py 2>&1 | grep gemm
for t in 128128 128160 256128 256160 064064; do for sk in 1 2 3; do LDX_GEMM_TILE=$t LDX_SPLITK=$sk python profiles/r06/gemm_sweep_probe.py 2>&1 | grep gemm; done; done ) | tee gpurun_out/r06l/gemm_sweep.txt
python profiles/r06/share_ab.py 20 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06l/share_ab.txt
