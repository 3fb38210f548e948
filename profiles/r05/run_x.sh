ROOT=$PWD; OUT=$ROOT/gpurun_out/r05x; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT -o bench --output-format csv -- python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary --no-graph > $OUT/bench_under_rocprof.json 2> $OUT/err.txt
python $ROOT/profiles/analyze_trace.py $(find $OUT -name "bench_kernel_trace.csv" | head -1) 45 > $OUT/forward_breakdown.txt 2>&1
find $OUT -name "bench_kernel_trace.csv" -delete
f=$(find $OUT -name "bench_kernel_stats.csv" | head -1); cp $f $OUT/bench_kernel_stats.csv 2>/dev/null
head -45 $OUT/forward_breakdown.txt | cut -c1-120; cut -c1-160 $OUT/bench_under_rocprof.json
