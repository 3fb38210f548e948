ROOT=$PWD; OUT=$ROOT/gpurun_out/r05pmc; mkdir -p $OUT; : > $OUT/pmc.txt
cd /tmp; export TMPDIR=/tmp
for set in "MfmaUtil VALUBusy" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VALU_TRANS SQ_WAIT_INST_LDS"; do
  rm -rf /tmp/pm
  rocprofv3 --pmc $set -d /tmp/pm -o p --output-format csv -- python $ROOT/profiles/kprobe.py attnsmall > /tmp/pm.log 2>&1
  python3 - >> $OUT/pmc.txt <<'PY'
import csv, glob, collections
fs = glob.glob("/tmp/pm/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for r in csv.DictReader(open(fs[0])):
    n = r["Kernel_Name"]
    if "attn" not in n: continue
    key = (n[:70], r.get("Grid_Size", r.get("Grid_Size_X", "")))
    acc[key][r["Counter_Name"]] += float(r["Counter_Value"]); disp[key].add(r["Dispatch_Id"])
for k, c in acc.items():
    print(k, {a: round(v / len(disp[k]), 1) for a, v in c.items()}, len(disp[k]))
PY
done
cat $OUT/pmc.txt
