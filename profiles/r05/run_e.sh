mkdir -p gpurun_out/r05e
for v in 0 42 162; do LDX_ATTN512_VAR=$v python profiles/attn512_probe.py 16384 2>&1 | grep attn512 | sed "s/$/ VAR=$v/" >> gpurun_out/r05e/var.txt; done
for abl in 1 2 4 6 7; do LDX_ATTN512_ABL=$abl python profiles/attn512_probe.py 16384 2>&1 | grep attn512 >> gpurun_out/r05e/var.txt; done
python profiles/attn512_probe.py 65536 4 2>&1 | grep attn512 >> gpurun_out/r05e/var.txt
for s in 1 2 3 4; do LDX_ATTN512_SPLITS=$s python profiles/attn512_probe.py 16384 2>&1 | grep attn512 >> gpurun_out/r05e/var.txt; done
cat gpurun_out/r05e/var.txt
python -m pytest tests/test_attn512_gpu.py tests/test_vae_clip_gpu.py -m gpu -x -q 2>&1 | tail -3
python profiles/vae_probe.py 128 2>&1 | head -5 > gpurun_out/r05e/vae128.txt; python profiles/vae_probe.py 256 2>&1 | head -5 > gpurun_out/r05e/vae256.txt; cat gpurun_out/r05e/vae128.txt gpurun_out/r05e/vae256.txt
