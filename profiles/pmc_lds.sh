#!/bin/bash
# LDS bank-conflict counters + time of one probe: profiles/pmc_lds.sh <kernel-substring> <python args...>
KSUB=$1; shift
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pm
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_LDS -d /tmp/pm -o p --output-format csv -- python $GRAFT_REPO_ROOT/"$@" > /tmp/pm.log 2>&1 || tail -3 /tmp/pm.log
python - "$KSUB" <<'PY'
import csv, glob, collections, sys
fs = glob.glob("/tmp/pm/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(float); disp = collections.defaultdict(set)
for r in csv.DictReader(open(fs[0])):
    if sys.argv[1] in r["Kernel_Name"]:
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); disp[r["Counter_Name"]].add(r["Dispatch_Id"])
print({k: round(v / max(len(disp[k]), 1)) for k, v in acc.items()})
PY
grep -E "attn|gemm|conv" /tmp/pm.log | head -4
