mkdir -p gpurun_out/r05r
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "conv3x3" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_esrgan_gpu.py tests/test_vae_clip_gpu.py tests/test_hires_gpu.py -m gpu -x -q 2>&1 | tail -2
python profiles/conv_patch_probe.py 0 5 2>&1 | grep conv; python profiles/conv_patch_probe.py 9 13 2>&1 | grep conv
timeout 300 python profiles/esrgan_probe.py 2>&1 | grep -E "RRDB|Upscale" | tee gpurun_out/r05r/esrgan.txt
