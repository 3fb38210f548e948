#!/bin/bash
cd /tmp; export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
for set in "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_EA0_RDREQ_sum TCC_BUBBLE_sum"; do
  rm -rf /tmp/pm
  rocprofv3 --pmc $set -d /tmp/pm -o p --output-format csv -- python $ROOT/profiles/kprobe.py conv1 > /tmp/pm.log 2>&1 || tail -3 /tmp/pm.log
  python - <<'PY'
import csv, glob, collections
fs = glob.glob("/tmp/pm/**/*counter_collection.csv", recursive=True)
if not fs: print("no counter file"); raise SystemExit
acc = collections.defaultdict(float); disp = set()
for r in csv.DictReader(open(fs[0])):
    if "gemm_kernel" in r["Kernel_Name"]:
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
n = max(len(disp), 1)
print({k: round(v / n) for k, v in acc.items()}, "dispatches", n)
PY
done
