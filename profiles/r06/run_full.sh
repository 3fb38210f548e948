# round 6: full evidence set of the current tree: smoke, the whole GPU suite, the default bench line, rocprofv3 trace + PMC passes
mkdir -p gpurun_out/r06full
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r06full/smoke.log 2>&1; tail -2 gpurun_out/r06full/smoke.log
timeout 3000 python -m pytest tests -m gpu -q -x --durations=15 > gpurun_out/r06full/gputests.log 2>&1; echo "gpu tests rc $?"; tail -25 gpurun_out/r06full/gputests.log
timeout 1200 python bench.py > gpurun_out/r06full/bench_default.json 2> gpurun_out/r06full/bench_default.err; echo "bench rc $?"; tail -c 3000 gpurun_out/r06full/bench_default.json
bash profiles/pmc_r06.sh > gpurun_out/r06full/pmc.log 2>&1; tail -5 gpurun_out/r06full/pmc.log
