#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel-trace stats of one Flux-dev-sized forward loop (profiles/flux_probe.py),
# 16-bit and MX fp8 mode.  Output -> gpurun_out/prof_flux_$1/ ; the kernel stats are what is kept.
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_flux_$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for mode in 0 1; do
  name=$([ $mode = 1 ] && echo mxfp8 || echo bf16)
  LDX_FLUX_FP8=$mode rocprofv3 --kernel-trace --stats -d $OUT -o flux_$name --output-format csv -- python $ROOT/profiles/flux_probe.py > $OUT/flux_${name}_probe.txt 2> $OUT/flux_${name}.err
  rm -f $OUT/flux_${name}_kernel_trace.csv
done
ls -la $OUT
