mkdir -p gpurun_out/r05n
ROOT=$PWD
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt && rocprofv3 --kernel-trace --stats -d /tmp/kt -o k --output-format csv -- python $ROOT/profiles/esrgan_probe.py quick > /tmp/kt.log 2>&1; f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1); head -12 "$f" > $ROOT/gpurun_out/r05n/esrgan_kernel_stats.csv;
  python3 - $(find /tmp/kt -name "*kernel_trace.csv" | head -1) > $ROOT/gpurun_out/r05n/esrgan_by_grid.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:60]; g = r.get("Grid_Size_X", r.get("Grid_Size", "?")) + "/lds" + r.get("LDS_Block_Size", r.get("LDS_Block_Size_v", "?"))
    acc[(k, g)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (k, g), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k:60s} grid {g:>9s} n={len(v):5d} avg {sum(v)/len(v):8.1f} us total {sum(v)/1e3:8.2f} ms")
PY
)
cat gpurun_out/r05n/esrgan_by_grid.txt | head -20
KERNEL=conv_patch_kernelIDF16bLi2 PROBE="python $ROOT/profiles/esrgan_probe.py quick" TAG=r05n bash profiles/pmc_attn512.sh > /dev/null 2>&1
KERNEL=conv_patch_kernelIDF16bLi4 PROBE="python $ROOT/profiles/esrgan_probe.py quick" TAG=r05n bash profiles/pmc_attn512.sh > /dev/null 2>&1
cat gpurun_out/r05n/pmc_*.txt
