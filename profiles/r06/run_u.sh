# the tile / split-K sweep of run_l.sh again, on the tree with the lean output stage (does the planner's choice still sit at the minimum?)
cd /root/repo
( python profiles/r06/gemm_sweep_probe.py 2>&1 | grep gemm
for t in 128128 128160 256128 256160 064064; do for sk in 1 2 3; do LDX_GEMM_TILE=$t LDX_SPLITK=$sk python profiles/r06/gemm_sweep_probe.py 2>&1 | grep gemm; done; done ) > gpurun_out/gemm_sweep_lean.txt
python - <<PY
import re,collections
best=collections.defaultdict(list)
for l in open("gpurun_out/gemm_sweep_lean.txt"):
    m=re.match(r"tile\s+(\S+) sk\s+(\S+) gemm (.*?):\s+([\d.]+) us",l)
    if m: best[m.group(3)].append((float(m.group(4)),m.group(1),m.group(2)))
for k,v in best.items():
    auto=[x for x in v if x[1]=="auto"][0]
    v.sort()
    print(k, "auto %.1f us | best %.1f us (tile %s sk %s) | next %.1f (%s %s)"%(auto[0],v[0][0],v[0][1],v[0][2],v[1][0],v[1][1],v[1][2]))
PY
