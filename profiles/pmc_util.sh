#!/bin/bash
# Run on the GPU box: rocprofv3 derived metrics (MfmaUtil, VALUBusy — gfx94x formulas, MI355X_MICROARCH.md) and raw
# SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE for the dominant kernels, one --pmc pass per set, no trace domains.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_util.txt
: > $OUT
cd /tmp; export TMPDIR=/tmp
run() {   # label, kernel substring, command...
  local label=$1 ksub=$2; shift 2
  for set in "MfmaUtil VALUBusy" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"; do
    rm -rf /tmp/pm
    rocprofv3 --pmc $set -d /tmp/pm -o p --output-format csv -- "$@" > /tmp/pm.log 2>&1 || tail -3 /tmp/pm.log >> $OUT
    python - "$label" "$ksub" >> $OUT <<'PY'
import csv, glob, collections, sys
fs = glob.glob("/tmp/pm/**/*counter_collection.csv", recursive=True)
if not fs:
    print(sys.argv[1], "no counter file"); sys.exit(0)
acc = collections.defaultdict(float); disp = collections.defaultdict(set)
for r in csv.DictReader(open(fs[0])):
    if sys.argv[2] in r["Kernel_Name"]:
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); disp[r["Counter_Name"]].add(r["Dispatch_Id"])
print(sys.argv[1], {k: round(v / max(len(disp[k]), 1), 2) for k, v in acc.items()}, "dispatches", max((len(d) for d in disp.values()), default=0))
PY
  done
}
run "attention D=40 B2 H8 N16384 (attn32ap_kernel, 8-wave two-group)" attn32ap_kernel python $ROOT/profiles/kprobe.py attn1
run "conv3x3 128^2 320->320 (gemm_pp_kernel MODE 1, 256x160 ping-pong tile)" gemm_pp_kernel python $ROOT/profiles/kprobe.py conv1
run "gemm bf16 8192^3 (gemm_pp_kernel MODE 0, 256x256 ping-pong tile)" gemm_pp_kernel python $ROOT/profiles/kprobe.py gemm1
run "MX fp8 GEMMs of profiles/mx_probe.py (gemm_pp_kernel<.., F8 = true>)" gemm_pp_kernel python $ROOT/profiles/mx_probe.py 2
cat $OUT
