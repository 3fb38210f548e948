set -x
mkdir -p gpurun_out/r05a
python -m pytest tests/test_step_cache_gpu.py tests/test_pingpong_gpu.py -k "64x160 or step_cache or context or emb_table or cfg_denoisers" -m gpu -x -q > gpurun_out/r05a/t1.log 2>&1; echo "t1 rc $?"
python -m pytest tests/test_fullwidth_gpu.py -m gpu -x -q -s > gpurun_out/r05a/t2.log 2>&1; echo "t2 rc $?"
python bench.py --no-configs --no-cpu-baseline > gpurun_out/r05a/b_new.json 2> gpurun_out/r05a/b_new.err; echo "b1 rc $?"
LDX_CTX_CACHE=0 LDX_EMB_TABLE=0 LDX_NO_TILE64X160=1 python bench.py --no-configs --no-cpu-baseline --no-secondary --no-parity-check > gpurun_out/r05a/b_old.json 2> gpurun_out/r05a/b_old.err; echo "b2 rc $?"
LDX_NO_TILE64X160=1 python bench.py --no-configs --no-cpu-baseline --no-secondary --no-parity-check > gpurun_out/r05a/b_cache_only.json 2> gpurun_out/r05a/b_cache.err; echo "b3 rc $?"
python profiles/shape_probe.py > gpurun_out/r05a/shape_probe.txt 2>&1
tail -5 gpurun_out/r05a/t1.log gpurun_out/r05a/t2.log
python - <<'PY'
import json
for n in ("b_new","b_old","b_cache_only"):
    try:
        d=json.loads([l for l in open(f"gpurun_out/r05a/{n}.json") if l.startswith("{")][0])
        print(n, d["value"], d["ms_per_step"], d["config"]["launches_per_step"], d.get("parity_check"), d["roofline"]["step_ms"])
    except Exception as e: print(n, "ERR", e)
PY
