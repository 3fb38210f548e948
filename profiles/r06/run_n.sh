mkdir -p gpurun_out/r06n
timeout 1500 python -m pytest tests/test_pingpong_gpu.py tests/test_engine_gpu.py -m gpu -q -k "ring or shared_cfg_prefix" 2>&1 | tail -4
python profiles/r06/gemm_sweep_probe.py 2>&1 | grep gemm | tee gpurun_out/r06n/gemm_auto.txt
python profiles/r06/share_ab.py 20 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06n/share_ab.txt
python profiles/shape_probe.py 64 bf16 2 cfg > gpurun_out/r06n/shape_64.txt 2>&1; head -8 gpurun_out/r06n/shape_64.txt | cut -c1-140
