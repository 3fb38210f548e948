mkdir -p gpurun_out/r06k
timeout 3000 python -m pytest tests -m gpu -q --durations=10 > gpurun_out/r06k/gputests.log 2>&1; echo "gpu tests rc $?"; tail -22 gpurun_out/r06k/gputests.log
python profiles/shape_probe.py 128 bf16 2 cfg > gpurun_out/r06k/shape_b2_cfg.txt 2>&1; head -3 gpurun_out/r06k/shape_b2_cfg.txt; grep rowgemm gpurun_out/r06k/shape_b2_cfg.txt | cut -c1-130
python profiles/r06/share_ab.py 20 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06k/share_ab.txt
