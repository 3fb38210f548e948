mkdir -p gpurun_out/r05k
LDX_ATTN_MX_QT=1 python -m pytest tests/test_attn_mx_gpu.py -m gpu -x -q 2>&1 | tail -2
python -m pytest tests/test_attn_mx_gpu.py -m gpu -x -q 2>&1 | tail -2
LDX_ATTN_MX_QT=1 python profiles/attn_mx_probe.py 2>&1 | tail -1 | sed 's/^/QT=1 /' | tee gpurun_out/r05k/probe.txt
python profiles/attn_mx_probe.py 2>&1 | tail -1 | sed 's/^/QT=2 /' | tee -a gpurun_out/r05k/probe.txt
