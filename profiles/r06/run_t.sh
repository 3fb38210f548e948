# tests touching the GEMM / reduce paths + bench (quick) on the current tree
cd /root/repo
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_pingpong_gpu.py tests/test_mx_gpu.py tests/test_rowgemm_gpu.py tests/test_engine_gpu.py tests/test_fullwidth_gpu.py tests/test_vae_gpu.py tests/test_esrgan_gpu.py -x -q -m gpu 2>&1 | tail -5
for rep in 1 2; do python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms/step', d['value'], 'it/s', 'parity', d['parity_check']['rel_l2'])"; done
