mkdir -p gpurun_out/r05h
python -m pytest tests/test_attn_mx_gpu.py -m gpu -x -q -s > gpurun_out/r05h/t_attn_mx.log 2>&1; echo "attn_mx rc $?"
grep -E "attention_fp8|flux shape|passed|failed|Error|assert" gpurun_out/r05h/t_attn_mx.log | head -30
python profiles/attn_mx_probe.py 2>&1 | tail -1 | tee gpurun_out/r05h/probe.txt
