mkdir -p gpurun_out/r05f
python -m pytest tests/test_pingpong_gpu.py -m gpu -x -q -k "ring" > gpurun_out/r05f/t_ring.log 2>&1; echo "ring tests rc $?"; tail -3 gpurun_out/r05f/t_ring.log
python profiles/shape_probe.py 2>&1 | grep -E "sum of|M2048 N1280 K1280|M2048 N3840|M512 N1280 K1280|M8192 N640|dispatches" > gpurun_out/r05f/shape_ring.txt
LDX_RING=0 python profiles/shape_probe.py 2>&1 | grep -E "sum of|M2048 N1280 K1280|M2048 N3840|M512 N1280 K1280|M8192 N640|dispatches" > gpurun_out/r05f/shape_noring.txt
LDX_RING=2 python profiles/shape_probe.py 2>&1 | grep -E "sum of|M2048 N1280 K1280|M2048 N3840|M512 N1280 K1280|M8192 N640|M2048 N1280 K5120|dispatches" > gpurun_out/r05f/shape_ring2.txt
head -20 gpurun_out/r05f/shape_ring.txt gpurun_out/r05f/shape_noring.txt gpurun_out/r05f/shape_ring2.txt
python bench.py --no-configs --no-cpu-baseline --no-secondary > gpurun_out/r05f/b_ring.json 2>/dev/null
LDX_RING=0 python bench.py --no-configs --no-cpu-baseline --no-secondary --no-parity-check > gpurun_out/r05f/b_noring.json 2>/dev/null
python - <<'PY'
import json
for n in ("b_ring","b_noring"):
    try:
        d=json.loads([l for l in open(f"gpurun_out/r05f/{n}.json") if l.startswith("{")][0])
        print(n, d["value"], d["ms_per_step"], d["config"]["launches_per_step"], (d.get("parity_check") or {}).get("rel_l2"), d["timing"]["region_ms_per_step"])
    except Exception as e: print(n, "ERR", e)
PY
