#!/bin/bash
# kernel-to-kernel gaps inside one graph-replayed step: rocprofv3 kernel trace of bench.py (graph mode), then for one forward
# in the timed region: sum of kernel durations vs wall span, and the distribution of the gaps between consecutive kernels.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/gp; rocprofv3 --kernel-trace -d /tmp/gp -o g --output-format csv -- python $ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline > /tmp/gp.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/gp/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "prep_image" in r["Kernel_Name"]]
s, e = idx[5], idx[6]          # a graph-replayed forward inside the timed region
fw = rows[s:e]
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000 for r in fw]
gaps = [(int(fw[i + 1]["Start_Timestamp"]) - int(fw[i]["End_Timestamp"])) / 1000 for i in range(len(fw) - 1)]
span = (int(fw[-1]["End_Timestamp"]) - int(fw[0]["Start_Timestamp"])) / 1000
gs = sorted(gaps)
print(f"{len(fw)} dispatches; sum of kernel time {sum(dur):.0f} us; span {span:.0f} us; total gap {sum(gaps):.0f} us "
      f"({100 * sum(gaps) / span:.1f} %); gap median {gs[len(gs) // 2]:.2f} us, p90 {gs[int(len(gs) * .9)]:.2f} us, max {gs[-1]:.1f} us; negative (overlap) {sum(1 for g in gaps if g < 0)}")
PY
