timeout 900 python -m pytest tests/test_rccl_gpu.py -m gpu -q 2>&1 | tail -30
