mkdir -p gpurun_out/r05v
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_engine_gpu.py -m gpu -x -q -k "conv3x3 or gemm or forward or apply_model or ksampler" 2>&1 | tail -2
timeout 600 python profiles/shape_probe.py > gpurun_out/r05v/shape_probe.txt 2>&1; grep -E "sum of op|sk11|sk3|sk6|sk2 " gpurun_out/r05v/shape_probe.txt | head -30
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | cut -c1-200
