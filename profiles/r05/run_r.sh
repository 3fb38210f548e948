mkdir -p gpurun_out/r05r
timeout 900 python -m pytest tests/test_vae_clip_gpu.py tests/test_hires_gpu.py tests/test_engine_gpu.py tests/test_fullwidth_gpu.py tests/test_pingpong_gpu.py -m gpu -x -q 2>&1 | tail -2
timeout 300 python profiles/vae_probe.py 128 2>&1 | head -24 | tee gpurun_out/r05r/vae_1024.txt
LDX_GN_FUSE_WIDE=0 timeout 300 python profiles/vae_probe.py 128 2>&1 | head -3
timeout 300 python profiles/vae_probe.py 256 2>&1 | head -3 | tee gpurun_out/r05r/vae_2048.txt
LDX_GN_FUSE_WIDE=0 timeout 300 python profiles/vae_probe.py 256 2>&1 | head -3
