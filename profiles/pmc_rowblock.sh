#!/bin/bash
# Run on the GPU box: MfmaUtil / VALUBusy and LDS / wait counters of the row-block kernels (profiles/kprobe.py rowblock), one --pmc pass per set.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_util_rowblock.txt
: > $OUT
cd /tmp; export TMPDIR=/tmp
python $ROOT/profiles/kprobe.py rowblock 2>&1 | grep -v amdgpu.ids >> $OUT
for set in "MfmaUtil VALUBusy" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/pm
  rocprofv3 --pmc $set -d /tmp/pm -o p --output-format csv -- python $ROOT/profiles/kprobe.py rowblock > /tmp/pm.log 2>&1 || tail -3 /tmp/pm.log >> $OUT
  python - >> $OUT <<'PY'
import csv, glob, collections
fs = glob.glob("/tmp/pm/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(float); disp = collections.defaultdict(set)
for r in csv.DictReader(open(fs[0])) if fs else []:
    n = r["Kernel_Name"]
    for k in ("rowgemm_kernel", "xattn_block_kernel", "ff_block_kernel"):
        if k in n:
            acc[(k, r["Counter_Name"])] += float(r["Counter_Value"]); disp[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
for (k, c), v in sorted(acc.items()):
    print(f"{k} {c} per-dispatch={v / len(disp[(k, c)]):.2f} dispatches={len(disp[(k, c)])}")
PY
done
cat $OUT
