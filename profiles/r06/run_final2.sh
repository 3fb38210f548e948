# round 6, FINAL tree (after the GEMM output-stage work): smoke, the whole GPU suite, the default bench line, rocprofv3 trace + PMC passes, per-shape tables
# (1024^2 CFG plan, CFG batch 16, 512^2, Flux fp8), the in-kernel stamps of the ping-pong tile's phases, and the same-box A/B of the lean output stage
mkdir -p gpurun_out/r06final2
O=gpurun_out/r06final2
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 3000 python -m pytest tests -m gpu -q --durations=10 > $O/gputests.log 2>&1; echo "gpu tests rc $?"; tail -16 $O/gputests.log
timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"; tail -c 1500 $O/bench_default.json
bash profiles/pmc_r06.sh > $O/pmc.log 2>&1; tail -3 $O/pmc.log | cut -c1-200
python profiles/shape_probe.py 128 bf16 2 cfg > $O/shape_probe_cfg.txt 2>&1; head -2 $O/shape_probe_cfg.txt | tail -1
python profiles/shape_probe.py 128 bf16 16 cfg > $O/shape_probe_b8_cfg.txt 2>&1; head -2 $O/shape_probe_b8_cfg.txt | tail -1
python profiles/shape_probe.py 64 bf16 2 cfg > $O/shape_probe_512.txt 2>&1; head -2 $O/shape_probe_512.txt | tail -1
LDX_FLUX_FP8=1 LDX_PROBE_SHAPES=1 python profiles/flux_probe.py 2>&1 | grep -v amdgpu.ids > $O/flux_shapes_fp8.txt; sed -n 4,8p $O/flux_shapes_fp8.txt
bash profiles/r06/run_q.sh > $O/pp_stamp_shapes.txt 2>&1; bash profiles/r06/run_r.sh > $O/pp_stamp_output_stage.txt 2>&1
LDX_EP_GENERAL=1 bash profiles/r06/run_r.sh > $O/pp_stamp_output_stage_general.txt 2>&1
bash profiles/r06/run_s.sh > $O/ep_ab.txt 2>&1; cat $O/ep_ab.txt
