# round 6, last tree (after the MX modulation weights): smoke, the whole GPU suite, the default bench line, the Flux fp8 per-shape table
mkdir -p gpurun_out/r06final3
O=gpurun_out/r06final3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 3000 python -m pytest tests -m gpu -q --durations=5 > $O/gputests.log 2>&1; echo "gpu tests rc $?"; tail -10 $O/gputests.log
timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"
python - <<PY
import json
d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["parity_check"]["rel_l2"], d["parity_check"]["ok"], d["roofline"]["frac"])
for k,v in d["secondary"].items():
    if isinstance(v,dict): print(k, {kk:vv for kk,vv in v.items() if kk in ("ms_per_step","it_per_s","image_steps_per_s","ms_per_evaluation","ms_per_forward_bs1","vae_decode_ms","e2e_s_per_image")})
PY
LDX_FLUX_FP8=1 LDX_PROBE_SHAPES=1 python profiles/flux_probe.py 2>&1 | grep -v amdgpu.ids > $O/flux_shapes_fp8.txt; sed -n 4,6p $O/flux_shapes_fp8.txt
