#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel-trace stats of bench.py + separate PMC passes (FETCH_SIZE /
# WRITE_SIZE, never combined with trace domains) on the dominant kernels.  Output -> gpurun_out/prof_$1/
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT -o bench --output-format csv -- python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary --no-graph > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
python $ROOT/profiles/analyze_trace.py $OUT/bench_kernel_trace.csv 30 > $OUT/forward_breakdown.txt 2>&1
rm -f $OUT/bench_kernel_trace.csv        # large; the stats + breakdown are what is kept
for what in attn1 conv1 gemm1; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pm
    rocprofv3 --pmc $ctr -d /tmp/pm -o p --output-format csv -- python $ROOT/profiles/kprobe.py $what > /tmp/pm.log 2>&1
    python - $what $ctr >> $OUT/pmc_traffic.txt <<'PY'
import csv, glob, collections, sys
fs = glob.glob("/tmp/pm/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: [0.0, set()])
for r in csv.DictReader(open(fs[0])):
    n = r["Kernel_Name"]
    if "ldx" in n and ("attn" in n or "gemm_kernel" in n or "gemm_pp_kernel" in n):
        a = acc[n]; a[0] += float(r["Counter_Value"]); a[1].add(r["Dispatch_Id"])
for n, (v, d) in acc.items():
    print(f"{sys.argv[1]} {sys.argv[2]} per-dispatch={v/len(d):.1f} dispatches={len(d)} kernel={n[:80]}")
PY
  done
done
cat $OUT/pmc_traffic.txt
