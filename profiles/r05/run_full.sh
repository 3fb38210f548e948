mkdir -p gpurun_out/r05full
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r05full/smoke.log 2>&1; tail -2 gpurun_out/r05full/smoke.log
timeout 3000 python -m pytest tests -m gpu -q -x --durations=15 > gpurun_out/r05full/gputests.log 2>&1; echo "gpu tests rc $?"; tail -25 gpurun_out/r05full/gputests.log
timeout 900 python bench.py > gpurun_out/r05full/bench_default.json 2> gpurun_out/r05full/bench_default.err; echo "bench rc $?"; tail -c 3000 gpurun_out/r05full/bench_default.json
