# row-block kernels vs tile GEMMs again, now that the tile GEMM's output stage is lean
cd /root/repo
for env in "X=0" "LDX_ROWGEMM640=0" "LDX_ROWGEMM=0" "X=0" "LDX_ROWGEMM640=0"; do
  env $env python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$env', d['ms_per_step'], 'ms/step', d['value'], 'it/s', 'launches', d['config'].get('launches_per_step'), 'parity', d['parity_check']['rel_l2'])"
done
