mkdir -p gpurun_out/r06h
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "geglu" 2>&1 | tail -4
for t in auto 256128 256256; do
  if [ $t = auto ]; then python profiles/r06/geglu_tile_probe.py 2>&1 | grep tile; else LDX_GEMM_TILE=$t python profiles/r06/geglu_tile_probe.py 2>&1 | grep tile; fi
done | tee gpurun_out/r06h/geglu_tiles.txt
python profiles/r06/share_ab.py 20 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06h/share_ab.txt
