# kernel durations vs the gaps between launches for the MX K-sweep (graph replays): python side prints wall per launch, the trace says how much of it is the kernel
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/ksw -o ksw --output-format csv -- python /root/repo/profiles/r06/mx_ksweep.py ${1:-3072} ${2:-512,3072} > /root/repo/gpurun_out/ksw.log 2>&1
cd /root/repo; grep "^M4352" gpurun_out/ksw.log
python - <<EOF
import csv,glob,statistics
f=glob.glob("/tmp/ksw/**/*kernel_trace.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "gemm_pp" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
d=[int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in rows]
g=[int(rows[i+1]["Start_Timestamp"])-int(rows[i]["End_Timestamp"]) for i in range(len(rows)-1)]+[0]
print(len(rows),"launches")
n=max(1,len(rows)//30)
for lo in range(0,len(rows),n):
    hi=min(len(rows),lo+n)
    print(lo, rows[lo]["Kernel_Name"][-70:], "grid", rows[lo]["Grid_Size_X"] if "Grid_Size_X" in rows[lo] else "", "dur med %.1f us"%(statistics.median(d[lo:hi])/1e3), "gap med %.1f us"%(statistics.median(g[lo:hi])/1e3))
EOF
