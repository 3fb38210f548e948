mkdir -p gpurun_out/r05d
for v in 0 42 82 122 162 41 161; do LDX_ATTN512_VAR=$v python profiles/attn512_probe.py 16384 2>&1 | grep attn512 | sed "s/$/ VAR=$v/" >> gpurun_out/r05d/var.txt; done
for abl in 1 2 4 6 7; do LDX_ATTN512_ABL=$abl python profiles/attn512_probe.py 16384 2>&1 | grep attn512 >> gpurun_out/r05d/var.txt; done
cat gpurun_out/r05d/var.txt
python -m pytest tests/test_attn512_gpu.py -m gpu -x -q 2>&1 | tail -2
