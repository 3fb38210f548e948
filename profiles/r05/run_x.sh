ROOT=$PWD; OUT=$ROOT/gpurun_out/r05x; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT -o bench --output-format csv -- python $ROOT/bench.py --latent 64 --steps 10 --warmup 2 --no-cpu-baseline --no-secondary --no-graph --no-parity-check > $OUT/bench_under_rocprof_512.json 2> $OUT/err.txt
python $ROOT/profiles/analyze_trace.py $(find $OUT -name "bench_kernel_trace.csv" | head -1) 60 > $OUT/forward_breakdown_512.txt 2>&1
find $OUT -name "bench_kernel_trace.csv" -delete
head -62 $OUT/forward_breakdown_512.txt | cut -c1-110; cut -c1-160 $OUT/bench_under_rocprof_512.json
