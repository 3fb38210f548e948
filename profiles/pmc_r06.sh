#!/bin/bash
# Run on the GPU box: round-6 evidence for the FINAL tree.
#  (1) rocprofv3 --kernel-trace --stats of the bench command (eager: every kernel its own dispatch) -> bench_kernel_stats.csv, forward_breakdown.txt
#  (2) PMC passes (one --pmc set per run, never combined with trace domains):
#      (a) the dominant kernel in isolation (profiles/kprobe.py attn1: attn40p at B2 H8 N16384 D40), as in rounds 4-5;
#      (b) IN SITU: every kernel of one eager CFG evaluation of the headline workload (profiles/r06/one_forward.py), averaged per (kernel, grid)
#          -> pmc_step_kernels.json: no carried-over row for anything that runs in the step.
# Output -> gpurun_out/prof_r06/
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_r06
mkdir -p $OUT; : > $OUT/pmc_kernels.txt; rm -f $OUT/pmc_step_kernels.json
cd /tmp; export TMPDIR=/tmp
if [ "$1" != "pmc-only" ]; then
rocprofv3 --kernel-trace --stats -d $OUT -o bench --output-format csv -- python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary --no-graph > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
python $ROOT/profiles/analyze_trace.py $(find $OUT -name "bench_kernel_trace.csv" | head -1) 40 > $OUT/forward_breakdown.txt 2>&1
find $OUT -name "bench_kernel_trace.csv" -delete
find $OUT -name "bench_kernel_stats.csv" -exec cp {} $OUT/bench_kernel_stats.csv \; 2>/dev/null
fi
SETS=("FETCH_SIZE" "WRITE_SIZE" "MfmaUtil VALUBusy" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_MFMA")
for set in "${SETS[@]}"; do
  rm -rf /tmp/pm
  rocprofv3 --pmc $set -d /tmp/pm -o p --output-format csv -- python $ROOT/profiles/kprobe.py attn1 > /tmp/pm.log 2>&1 || tail -3 /tmp/pm.log >> $OUT/pmc_kernels.txt
  SUB=attn40p python3 - attn1 >> $OUT/pmc_kernels.txt <<'PY'
import csv, glob, collections, sys, os
fs = glob.glob("/tmp/pm/**/*counter_collection.csv", recursive=True)
if not fs:
    print(sys.argv[1], "no counter file"); sys.exit(0)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for r in csv.DictReader(open(fs[0])):
    n = r["Kernel_Name"]
    if os.environ["SUB"] not in n: continue
    acc[n][r["Counter_Name"]] += float(r["Counter_Value"]); disp[n].add(r["Dispatch_Id"])
for n, c in acc.items():
    print(sys.argv[1], {k: round(v / len(disp[n]), 2) for k, v in c.items()}, "dispatches", len(disp[n]), "kernel", n[:110])
PY
  rm -rf /tmp/pm
  rocprofv3 --pmc $set -d /tmp/pm -o p --output-format csv -- python $ROOT/profiles/r06/one_forward.py 2 > /tmp/pm.log 2>&1 || tail -3 /tmp/pm.log >> $OUT/pmc_kernels.txt
  f=$(find /tmp/pm -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 $ROOT/profiles/r06/pmc_aggregate.py $f $OUT/pmc_step_kernels.json >> $OUT/pmc_kernels.txt 2>&1; else echo "in-situ pass '$set': no counter file" >> $OUT/pmc_kernels.txt; tail -3 /tmp/pm.log >> $OUT/pmc_kernels.txt; fi
done
cat $OUT/pmc_kernels.txt | cut -c1-300
