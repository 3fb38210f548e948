# round 6, final tree: smoke, the whole GPU suite, the default bench line, rocprofv3 trace + PMC passes, per-shape tables (1024^2 CFG plan, CFG batch 16, 512^2)
mkdir -p gpurun_out/r06final
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r06final/smoke.log 2>&1; tail -2 gpurun_out/r06final/smoke.log
timeout 3000 python -m pytest tests -m gpu -q --durations=10 > gpurun_out/r06final/gputests.log 2>&1; echo "gpu tests rc $?"; tail -16 gpurun_out/r06final/gputests.log
timeout 1500 python bench.py > gpurun_out/r06final/bench_default.json 2> gpurun_out/r06final/bench_default.err; echo "bench rc $?"; tail -c 1500 gpurun_out/r06final/bench_default.json
bash profiles/pmc_r06.sh > gpurun_out/r06final/pmc.log 2>&1; tail -3 gpurun_out/r06final/pmc.log | cut -c1-200
python profiles/shape_probe.py 128 bf16 2 cfg > gpurun_out/r06final/shape_probe_cfg.txt 2>&1; head -2 gpurun_out/r06final/shape_probe_cfg.txt | tail -1
python profiles/shape_probe.py 128 bf16 16 cfg > gpurun_out/r06final/shape_probe_b8_cfg.txt 2>&1; head -2 gpurun_out/r06final/shape_probe_b8_cfg.txt | tail -1
python profiles/shape_probe.py 64 bf16 2 cfg > gpurun_out/r06final/shape_probe_512.txt 2>&1; head -2 gpurun_out/r06final/shape_probe_512.txt | tail -1
