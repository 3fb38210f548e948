# CFG batch 16: all K = 640 row-block GEMMs off (LDX_ROWGEMM640=0: LayerNorm / GroupNorm launches + tile GEMMs) against the default (plain ones on tile GEMMs above M 16384)
cd /root/repo
for e in "X=1" "LDX_ROWGEMM640=0" "X=1" "LDX_ROWGEMM640=0"; do
  env $e python bench.py --batch 8 --steps 10 --warmup 2 --no-cpu-baseline --no-secondary --no-parity-check 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$e', d['ms_per_step'], 'ms/step', 'launches', d['config'].get('launches_per_step'))"
done
