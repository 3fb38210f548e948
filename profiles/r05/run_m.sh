mkdir -p gpurun_out/r05m
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "conv3x3" > gpurun_out/r05m/t_conv.log 2>&1; echo "conv tests rc $?"
tail -5 gpurun_out/r05m/t_conv.log
timeout 900 python -m pytest tests/test_esrgan_gpu.py -m gpu -x -q -s > gpurun_out/r05m/t_esrgan.log 2>&1; echo "esrgan tests rc $?"
grep -E "rel|passed|failed|Error|assert" gpurun_out/r05m/t_esrgan.log | head -20
timeout 300 python profiles/esrgan_probe.py > gpurun_out/r05m/esrgan_patch.txt 2>&1; cat gpurun_out/r05m/esrgan_patch.txt
LDX_CONV_PATCH=0 timeout 300 python profiles/esrgan_probe.py > gpurun_out/r05m/esrgan_old.txt 2>&1; cat gpurun_out/r05m/esrgan_old.txt
LDX_FLUX_FP8=1 timeout 300 python profiles/flux_probe.py > gpurun_out/r05m/flux_fp8_qt1.txt 2>&1
grep -E "Flux DiT forward|attn|rope|vt_quant" gpurun_out/r05m/flux_fp8_qt1.txt | head -8
