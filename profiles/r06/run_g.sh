mkdir -p gpurun_out/r06g
python profiles/r06/split_chain_probe.py 128 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06g/split_chain.txt
python profiles/r06/split_chain_probe.py 64 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r06g/split_chain.txt
