mkdir -p gpurun_out/r05o
python profiles/conv_patch_probe.py 2>&1 | grep conv | tee gpurun_out/r05o/patch.txt
LDX_CONV_PATCH=0 python profiles/conv_patch_probe.py 2>&1 | grep conv | tee gpurun_out/r05o/igemm.txt
