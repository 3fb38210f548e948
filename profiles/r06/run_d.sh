mkdir -p gpurun_out/r06d
timeout 1500 python -m pytest tests/test_timestep_gpu.py tests/test_engine_gpu.py tests/test_step_cache_gpu.py -m gpu -q 2>&1 | tail -8
for t in auto 128160 256160 256128 128128 64064 128064; do
  if [ $t = auto ]; then python profiles/r06/conv_tile_probe.py 2>&1 | grep tile; else LDX_GEMM_TILE=$t python profiles/r06/conv_tile_probe.py 2>&1 | grep tile; fi
done | tee gpurun_out/r06d/conv_tiles.txt
python profiles/r06/share_ab.py 20 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06d/share_ab.txt
python profiles/shape_probe.py 128 bf16 2 cfg > gpurun_out/r06d/shape_b2_cfg.txt 2>&1; head -3 gpurun_out/r06d/shape_b2_cfg.txt
