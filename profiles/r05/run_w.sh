echo "with bias + residual"; python profiles/kprobe.py small 2>&1 | grep "^gemm"
echo "no epilogue operands"; NOEPI=1 python profiles/kprobe.py small 2>&1 | grep "^gemm"
