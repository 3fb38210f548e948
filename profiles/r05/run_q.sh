ROOT=$PWD
mkdir -p gpurun_out/r05q
cd /tmp && export TMPDIR=/tmp
for a in 0 4 16 23 31; do
  rm -rf /tmp/kt
  LDX_CP_ABL=$a rocprofv3 --kernel-trace --stats -d /tmp/kt -o k --output-format csv -- python $ROOT/profiles/conv_patch_probe.py 2 > /tmp/kt.log 2>&1
  f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1)
  echo "ABL=$a $(grep conv_patch $f | head -1 | cut -c1-200)"
done | tee $ROOT/gpurun_out/r05q/abl_gpu_times.txt
