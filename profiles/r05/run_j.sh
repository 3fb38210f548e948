mkdir -p gpurun_out/r05j
python -m pytest tests/test_attn_mx_gpu.py -m gpu -x -q 2>&1 | tail -2
python profiles/attn_mx_probe.py 2>&1 | tail -1 | tee gpurun_out/r05j/probe.txt
TAG=r05j KERNEL=attn_mx_kernel PROBE="python $GRAFT_REPO_ROOT/profiles/attn_mx_probe.py" bash profiles/pmc_attn512.sh 2>&1 | tail -8
