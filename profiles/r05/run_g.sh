mkdir -p gpurun_out/r05g
python -m pytest tests/test_attn_mx_gpu.py -m gpu -x -q -s > gpurun_out/r05g/t_attn_mx.log 2>&1; echo "attn_mx rc $?"
grep -E "attention_fp8|rope_mx|flux shape|passed|failed|Error|assert" gpurun_out/r05g/t_attn_mx.log | head -40
