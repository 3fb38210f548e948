timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_mx_gpu.py tests/test_attn_mx_gpu.py tests/test_flux_gpu.py -m gpu -x -q 2>&1 | tail -2
timeout 300 python profiles/ln_probe.py 2>&1 | tail -5
LDX_FLUX_FP8=1 timeout 400 python profiles/flux_probe.py 2>&1 | grep -E "Flux DiT forward|ln_kernel|attn_mx|gemm_kernel<mxfp8|vt_quant|rope"
timeout 400 python profiles/flux_probe.py 2>&1 | grep -E "Flux DiT forward|ln_kernel"
