#!/bin/bash
# Run on the GPU box: PMC counters of the three D = 40 attention kernels on B2 H8 N16384 (stand-alone harness, one --pmc pass per set, no trace domains).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_attn_pipe.txt
: > $OUT
cd /tmp; export TMPDIR=/tmp
if [ -n "$1" ]; then rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u | tr '\n' ' ' > $ROOT/gpurun_out/sq_counters.txt; fi
for which in ${WHICH:-0 1}; do
  for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU" "SQ_INSTS_MFMA SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA"; do
    rm -rf /tmp/pm
    rocprofv3 --pmc $set -d /tmp/pm -o p --output-format csv -- $ROOT/profiles/ubench/attn_pipe_test ${NBIG:-16384} 4 $which ${DD:-40} > /tmp/pm.log 2>&1 || tail -3 /tmp/pm.log >> $OUT
    python3 - "$which" >> $OUT <<'PY'
import csv, glob, collections, sys
fs = glob.glob("/tmp/pm/**/*counter_collection.csv", recursive=True)
if not fs:
    print(sys.argv[1], "no counter file"); sys.exit(0)
acc = collections.defaultdict(float); disp = collections.defaultdict(set); name = ""
for r in csv.DictReader(open(fs[0])):
    if "attn" in r["Kernel_Name"] and "knorm" not in r["Kernel_Name"]:
        name = r["Kernel_Name"][:40]
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); disp[r["Counter_Name"]].add(r["Dispatch_Id"])
print("variant", sys.argv[1], name, {k: round(v / max(len(disp[k]), 1)) for k, v in acc.items()}, "dispatches", max((len(d) for d in disp.values()), default=0))
PY
  done
done
cat $OUT
