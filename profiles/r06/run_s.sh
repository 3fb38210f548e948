# same-box A/B of the GEMM output stage: general stage for every epilogue (LDX_EP_GENERAL=1) vs the lean paths (default)
cd /root/repo
for rep in 1 2; do
  for g in 1 0; do
    LDX_EP_GENERAL=$g python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('LDX_EP_GENERAL=$g', d['ms_per_step'], 'ms/step', d['value'], 'it/s', 'parity', d['parity_check']['rel_l2'])"
  done
done
for g in 1 0; do
  LDX_EP_GENERAL=$g LDX_FLUX_FP8=1 python profiles/flux_probe.py 2>&1 | grep "Flux DiT forward" | sed "s/^/LDX_EP_GENERAL=$g /"
done
