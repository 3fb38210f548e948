ROOT=$PWD; OUT=$ROOT/gpurun_out/r05pmc; mkdir -p $OUT; : > $OUT/pmc_conv1.txt
cd /tmp; export TMPDIR=/tmp
for set in "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  rm -rf /tmp/pm
  rocprofv3 --pmc $set -d /tmp/pm -o p --output-format csv -- python $ROOT/profiles/kprobe.py conv1 > /tmp/pm.log 2>&1 || tail -2 /tmp/pm.log >> $OUT/pmc_conv1.txt
  python3 - >> $OUT/pmc_conv1.txt <<'PY'
import csv, glob, collections
fs = glob.glob("/tmp/pm/**/*counter_collection.csv", recursive=True)
if fs:
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
    for r in csv.DictReader(open(fs[0])):
        n = r["Kernel_Name"]
        if "gemm_pp" not in n: continue
        acc[n[:60]][r["Counter_Name"]] += float(r["Counter_Value"]); disp[n[:60]].add(r["Dispatch_Id"])
    for k, c in acc.items():
        print(k, {a: round(v / len(disp[k]), 1) for a, v in c.items()}, len(disp[k]))
else:
    print("no counter file")
PY
done
cat $OUT/pmc_conv1.txt
