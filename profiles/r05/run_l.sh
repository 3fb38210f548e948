mkdir -p gpurun_out/r05l
python -m pytest tests/test_flux_gpu.py tests/test_mx_gpu.py -m gpu -x -q -s > gpurun_out/r05l/t_flux.log 2>&1; echo "flux tests rc $?"
grep -E "Flux MX fp8|fp8 \+ FBCache|passed|failed|Error|assert" gpurun_out/r05l/t_flux.log | head -20
LDX_FLUX_FP8=1 python profiles/flux_probe.py > gpurun_out/r05l/flux_fp8_attn8.txt 2>&1
LDX_FLUX_FP8=1 LDX_FLUX_FP8_ATTN=0 python profiles/flux_probe.py > gpurun_out/r05l/flux_fp8_attn16.txt 2>&1
grep -E "Flux DiT forward|attn|rope|vt_quant" gpurun_out/r05l/flux_fp8_attn8.txt | head -8
grep -E "Flux DiT forward|attn|rope" gpurun_out/r05l/flux_fp8_attn16.txt | head -8
