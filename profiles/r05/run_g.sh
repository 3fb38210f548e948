timeout 900 python -m pytest tests/test_pingpong_gpu.py tests/test_ops_gpu.py -m gpu -x -q -k "forced or geglu" 2>&1 | tail -2
echo default; python profiles/kprobe.py geglu 2>&1 | grep "^gemm"
echo forced256256; LDX_GEMM_TILE=256256 python profiles/kprobe.py geglu 2>&1 | grep "^gemm"
echo forced256128; LDX_GEMM_TILE=256128 python profiles/kprobe.py geglu 2>&1 | grep "^gemm"
