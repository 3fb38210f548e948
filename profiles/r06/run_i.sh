mkdir -p gpurun_out/r06i
timeout 1800 python -m pytest tests/test_splitk_gn_gpu.py tests/test_fullwidth_gpu.py tests/test_ops_gpu.py -m gpu -q -x 2>&1 | tail -12
python profiles/shape_probe.py 128 bf16 2 cfg > gpurun_out/r06i/shape_b2_cfg.txt 2>&1; head -3 gpurun_out/r06i/shape_b2_cfg.txt; grep -E "splitk_gn|gn_|sk[0-9]" gpurun_out/r06i/shape_b2_cfg.txt | cut -c1-140 | head -50
for m in 1 0; do LDX_SKGN=$m python profiles/r06/share_ab.py 20 2>&1 | grep -v amdgpu.ids | sed "s/^/SKGN=$m /"; done | tee gpurun_out/r06i/skgn_ab.txt
