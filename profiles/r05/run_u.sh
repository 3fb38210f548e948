timeout 900 python -m pytest tests/test_pingpong_gpu.py tests/test_ops_gpu.py -m gpu -x -q -k "gemm or ring" 2>&1 | tail -2
echo "RING64=1"; python profiles/kprobe.py small 2>&1 | grep "^gemm"
echo "RING64=0"; LDX_RING64=0 python profiles/kprobe.py small 2>&1 | grep "^gemm"
