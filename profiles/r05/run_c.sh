set -x
mkdir -p gpurun_out/r05c
for abl in 0 1 2 4 6 7; do LDX_ATTN512_ABL=$abl python profiles/attn512_probe.py 16384 >> gpurun_out/r05c/abl.txt 2>&1; done
LDX_ATTN512_SPLITS=1 python profiles/attn512_probe.py 16384 >> gpurun_out/r05c/abl.txt 2>&1
python profiles/attn512_probe.py 65536 4 >> gpurun_out/r05c/abl.txt 2>&1
grep attn512 gpurun_out/r05c/abl.txt
TAG=r05c bash profiles/pmc_attn512.sh
