mkdir -p gpurun_out/r06e
timeout 1500 python -m pytest tests/test_timestep_gpu.py tests/test_engine_gpu.py tests/test_step_cache_gpu.py tests/test_ops_gpu.py tests/test_fullwidth_gpu.py -m gpu -q 2>&1 | tail -8
for sk in 1 2; do LDX_SPLITK=$sk LDX_GEMM_TILE=128160 python profiles/r06/conv_tile_probe.py 2>&1 | grep tile | sed "s/^/splitk $sk /"; done | tee gpurun_out/r06e/conv_b1_splitk.txt
python profiles/r06/share_ab.py 20 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06e/share_ab.txt
python profiles/shape_probe.py 128 bf16 2 cfg > gpurun_out/r06e/shape_b2_cfg.txt 2>&1; head -3 gpurun_out/r06e/shape_b2_cfg.txt
timeout 900 python bench.py --no-configs > gpurun_out/r06e/bench_noconfigs.json 2> gpurun_out/r06e/bench_noconfigs.err; echo "bench rc $?"; python -c "
import json; d=json.loads(open('gpurun_out/r06e/bench_noconfigs.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['step_frac'], d['parity_check'])"
