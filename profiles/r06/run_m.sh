timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -q -k "shared_cfg_prefix or denoise_cfg" 2>&1 | tail -12
