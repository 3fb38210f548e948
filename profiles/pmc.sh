#!/bin/bash
# usage: profiles/pmc.sh <kernel-substring> <python args...>   (run on the GPU box; separate --pmc passes)
KSUB=$1; shift
cd /tmp; export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_SALU SQ_WAVES SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_SCA"; do
  rm -rf /tmp/pm
  rocprofv3 --pmc $set -d /tmp/pm -o p --output-format csv -- python "$@" > /tmp/pm.log 2>&1 || tail -5 /tmp/pm.log
  python - "$KSUB" <<'PY'
import csv, glob, collections, sys
fs = glob.glob("/tmp/pm/**/*counter_collection.csv", recursive=True)
if not fs:
    print("no counter file", glob.glob("/tmp/pm/**", recursive=True)[:10]); sys.exit(0)
acc = collections.defaultdict(float); disp = set()
for r in csv.DictReader(open(fs[0])):
    if sys.argv[1] in r["Kernel_Name"]:
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
n = max(len(disp), 1)
print({k: round(v / n) for k, v in acc.items()}, "dispatches", n)
PY
done
