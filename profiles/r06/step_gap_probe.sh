#!/bin/bash
# Gaps BETWEEN consecutive graph-replayed steps (the hipGraph launch, the fill launch and the sampler-step launch sit there): rocprofv3 kernel trace of bench.py in
# graph mode; for the timed region: span of each step (first prep_image start to the next one) against the sum of its kernel times, and the idle time around the step seam.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/gp; rocprofv3 --kernel-trace -d /tmp/gp -o g --output-format csv -- python $ROOT/bench.py --steps 8 --warmup 3 --repeats 1 --no-cpu-baseline --no-secondary --no-parity-check > /tmp/gp.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/gp/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "prep_image" in r["Kernel_Name"]]
for k in range(5, min(10, len(idx) - 1)):
    s, e = idx[k], idx[k + 1]
    fw = rows[s:e]
    dur = sum((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in fw) / 1000
    span = (int(rows[e]["Start_Timestamp"]) - int(fw[0]["Start_Timestamp"])) / 1000
    gaps = [(int(fw[i + 1]["Start_Timestamp"]) - int(fw[i]["End_Timestamp"])) / 1000 for i in range(len(fw) - 1)]
    seam = (int(rows[e]["Start_Timestamp"]) - int(fw[-1]["End_Timestamp"])) / 1000
    big = sorted(((g, fw[i]["Kernel_Name"][:40], fw[i + 1]["Kernel_Name"][:40]) for i, g in enumerate(gaps)), reverse=True)[:4]
    print(f"step {k}: {len(fw)} dispatches, kernel time {dur:.0f} us, span to the next step's first kernel {span:.0f} us, idle {span - dur:.0f} us (seam after the last kernel {seam:.1f} us); largest gaps {[(round(g, 1), a, b) for g, a, b in big]}")
PY
