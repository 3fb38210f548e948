#!/bin/bash
# Per-kernel-name totals of a few SQ counters over a short eager bench run (one --pmc pass, no trace domains).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pm
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_LDS -d /tmp/pm -o p --output-format csv -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph > /tmp/pm.log 2>&1 || tail -3 /tmp/pm.log
python - <<'PY'
import csv, glob, collections
fs = glob.glob("/tmp/pm/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(fs[0])):
    acc[r["Kernel_Name"][:70]][r["Counter_Name"]] += float(r["Counter_Value"])
rows = sorted(acc.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"])
for n, c in rows[:22]:
    print(f"{n:70s} wave_cyc {c['SQ_WAVE_CYCLES']:.3e} lds_active {c['SQ_ACTIVE_INST_LDS']:.3e} bank_conflict {c['SQ_LDS_BANK_CONFLICT']:.3e} wait_lds {c['SQ_WAIT_INST_LDS']:.3e}")
PY
