#!/bin/bash
# Run on the GPU box: per-kernel PMC evidence for round 4 — HBM traffic (FETCH_SIZE / WRITE_SIZE, one counter per pass) and utilisation
# (MfmaUtil VALUBusy; SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA; SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY) of
# the kernel classes VERDICT r03 asked for.  --pmc passes only, never combined with trace domains.  Output -> gpurun_out/pmc_r04.txt
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_r04.txt
: > $OUT
cd /tmp; export TMPDIR=/tmp
for what in attn1 attn128 conv1 gemm1 gemm640 deep1 rowblock; do
  for set in FETCH_SIZE WRITE_SIZE "MfmaUtil VALUBusy" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
    rm -rf /tmp/pm
    rocprofv3 --pmc $set -d /tmp/pm -o p --output-format csv -- python $ROOT/profiles/kprobe.py $what > /tmp/pm.log 2>&1 || tail -3 /tmp/pm.log >> $OUT
    python - $what >> $OUT <<'PY'
import csv, glob, collections, sys
fs = glob.glob("/tmp/pm/**/*counter_collection.csv", recursive=True)
if not fs:
    print(sys.argv[1], "no counter file"); sys.exit(0)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for r in csv.DictReader(open(fs[0])):
    n = r["Kernel_Name"]
    if "ldx" not in n: continue
    acc[n][r["Counter_Name"]] += float(r["Counter_Value"]); disp[n].add(r["Dispatch_Id"])
for n, c in acc.items():
    print(sys.argv[1], {k: round(v / len(disp[n]), 2) for k, v in c.items()}, "dispatches", len(disp[n]), "kernel", n[:110])
PY
  done
done
cat $OUT
