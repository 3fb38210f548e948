#!/bin/bash
# Run on the GPU box: PMC counters of one kernel (KERNEL=<substring of its name>, PROBE=<command that launches it>; default attn512_kernel at N = 16384);
# one --pmc pass per set, no trace domains.
export KERNEL
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/${TAG:-r05c}/pmc_${KERNEL:-attn512_kernel}.txt
mkdir -p $(dirname $OUT); : > $OUT
cd /tmp; export TMPDIR=/tmp
for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU" "SQ_INSTS_MFMA SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/pm
  rocprofv3 --pmc $set -d /tmp/pm -o p --output-format csv -- ${PROBE:-python $ROOT/profiles/attn512_probe.py ${NBIG:-16384} 4} > /tmp/pm.log 2>&1 || tail -3 /tmp/pm.log >> $OUT
  python3 - >> $OUT <<'PY'
import csv, glob, collections, os
fs = glob.glob("/tmp/pm/**/*counter_collection.csv", recursive=True)
if not fs:
    print("no counter file")
else:
    acc = collections.defaultdict(float); disp = collections.defaultdict(set)
    for r in csv.DictReader(open(fs[0])):
        if os.environ.get("KERNEL", "attn512_kernel") in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); disp[r["Counter_Name"]].add(r["Dispatch_Id"])
    print({k: round(v / max(len(disp[k]), 1)) for k, v in acc.items()}, "dispatches", max((len(d) for d in disp.values()), default=0))
PY
done
cat $OUT
