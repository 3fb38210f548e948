#!/bin/bash
# usage: profiles/pmc2.sh <kernel-substring> "<counters>" <python args...>
KSUB=$1; CTRS=$2; shift; shift
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pm
rocprofv3 --pmc $CTRS -d /tmp/pm -o p --output-format csv -- python "$@" > /tmp/pm.log 2>&1 || tail -5 /tmp/pm.log
python - "$KSUB" <<'PY'
import csv, glob, collections, sys
fs = glob.glob("/tmp/pm/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(float); disp = set()
for r in csv.DictReader(open(fs[0])):
    if sys.argv[1] in r["Kernel_Name"]:
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
n = max(len(disp), 1)
print({k: round(v / n) for k, v in acc.items()}, "dispatches", n)
PY
