run() { env "$@" python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --no-parity-check 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'])"; }
for cfg in "A=0" "LDX_RING=0" "LDX_RING=2" "LDX_NO_SPLIT2=1" "LDX_PP_MINK=512" "LDX_PP_MINK=2048" "LDX_GN_FUSE_SPLITK=0" "LDX_LNFOLD=1" "LDX_NO_TILE160=1" "LDX_ATTN_KPF=0" "A=0"; do
  echo "$cfg $(run $cfg) $(run $cfg)"
done
