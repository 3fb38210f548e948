#!/bin/bash
# Run on the GPU box: round-5 evidence.  (1) rocprofv3 --kernel-trace --stats of the bench command (eager, so that every kernel is a separate dispatch);
# (2) PMC passes (one --pmc set per run, never combined with trace domains) for the dominant kernel (attn1) and the round's new kernels.
# Output -> gpurun_out/prof_r05/
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_r05
mkdir -p $OUT; : > $OUT/pmc_kernels.txt
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT -o bench --output-format csv -- python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary --no-graph > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
python $ROOT/profiles/analyze_trace.py $(find $OUT -name "bench_kernel_trace.csv" | head -1) 30 > $OUT/forward_breakdown.txt 2>&1
find $OUT -name "bench_kernel_trace.csv" -delete
find $OUT -name "bench_kernel_stats.csv" -exec cp {} $OUT/bench_kernel_stats.csv \; 2>/dev/null
pmc() {   # tag, kernel-name substring, command...
  local tag=$1 sub=$2; shift 2
  for set in FETCH_SIZE WRITE_SIZE "MfmaUtil VALUBusy" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA"; do
    rm -rf /tmp/pm
    rocprofv3 --pmc $set -d /tmp/pm -o p --output-format csv -- "$@" > /tmp/pm.log 2>&1 || tail -3 /tmp/pm.log >> $OUT/pmc_kernels.txt
    SUB="$sub" python3 - $tag >> $OUT/pmc_kernels.txt <<'PY'
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
  done
}
pmc attn1 attn40p python $ROOT/profiles/kprobe.py attn1
pmc attn512_N16384 attn512_kernel python $ROOT/profiles/attn512_probe.py 16384 4
pmc attn_mx_4352 attn_mx_kernel python $ROOT/profiles/attn_mx_probe.py
pmc conv_patch_1024_128to128 conv_patch python $ROOT/profiles/conv_patch_probe.py 6
pmc conv_patch_512_192to32 conv_patch python $ROOT/profiles/conv_patch_probe.py 2
pmc conv_patch_512_192to64 conv_patch python $ROOT/profiles/conv_patch_probe.py 3
cat $OUT/pmc_kernels.txt
