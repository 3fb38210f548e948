python profiles/kprobe.py geglu 2>&1 | grep "^gemm"
timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_fullwidth_gpu.py -m gpu -x -q 2>&1 | tail -2
for i in 1 2 3; do
  (cd _ab && python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --no-parity-check 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('old', d['ms_per_step'], d['value'])")
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --no-parity-check 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('new', d['ms_per_step'], d['value'])"
done
