mkdir -p gpurun_out/r05r
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "conv3x3" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_esrgan_gpu.py tests/test_vae_clip_gpu.py tests/test_hires_gpu.py -m gpu -x -q 2>&1 | tail -2
timeout 300 python profiles/esrgan_probe.py 2>&1 | grep -E "RRDB|Upscale" | tee gpurun_out/r05r/esrgan.txt
timeout 300 python profiles/vae_probe.py 128 2>&1 | head -8 | tee gpurun_out/r05r/vae_1024.txt
timeout 300 python profiles/vae_probe.py 256 2>&1 | head -8 | tee gpurun_out/r05r/vae_2048.txt
