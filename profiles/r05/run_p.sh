timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "conv3x3_patch" 2>&1 | tail -3
timeout 300 python profiles/conv_patch_probe.py 2>&1 | grep conv
for a in 4 16 23 31; do echo "ABL=$a"; LDX_CP_ABL=$a timeout 120 python profiles/conv_patch_probe.py 2 2>&1 | grep "conv"; done
timeout 300 python profiles/esrgan_probe.py quick 2>&1 | grep RRDB
