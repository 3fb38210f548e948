# CFG batch 16 (config 3's per-GPU shard): plain K = 640 row-block GEMMs vs tile GEMMs (LDX_ROWGEMM_PLAIN640_MAXM: 1000000000 = row block always, 16384 = new default)
cd /root/repo
for m in 1000000000 16384 1000000000 16384; do
  LDX_ROWGEMM_PLAIN640_MAXM=$m python bench.py --batch 8 --steps 10 --warmup 2 --no-cpu-baseline --no-secondary --no-parity-check 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('LDX_ROWGEMM_PLAIN640_MAXM=$m', d['ms_per_step'], 'ms/step', d['value'], d['unit'], 'launches', d['config'].get('launches_per_step'))"
done
