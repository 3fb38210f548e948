mkdir -p gpurun_out/r06f
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_engine_gpu.py tests/test_vae_clip_gpu.py -m gpu -q 2>&1 | tail -4
python profiles/shape_probe.py 128 bf16 2 cfg > gpurun_out/r06f/shape_b2_cfg.txt 2>&1; head -3 gpurun_out/r06f/shape_b2_cfg.txt; grep "gn_" gpurun_out/r06f/shape_b2_cfg.txt | cut -c1-130
python profiles/r06/two_chain_probe.py 20 128 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06f/two_chain.txt
python profiles/r06/two_chain_probe.py 40 64 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r06f/two_chain.txt
