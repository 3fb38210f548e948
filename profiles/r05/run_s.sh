python profiles/kprobe.py rowblock 2>&1 | grep -E "rowgemm|xattn|ff_block"
python -m pytest tests/test_rowblock_gpu.py tests/test_ops_gpu.py -m gpu -x -q -k "rowgemm or xattn or ff_block or rowblock" 2>&1 | tail -2
