set -x
mkdir -p gpurun_out/r05b
python -m pytest tests/test_attn512_gpu.py -m gpu -x -q -s > gpurun_out/r05b/t_attn512.log 2>&1; echo "attn512 rc $?"
python -m pytest tests/test_vae_clip_gpu.py tests/test_step_cache_gpu.py tests/test_attn_pipe_gpu.py -m gpu -x -q > gpurun_out/r05b/t_misc.log 2>&1; echo "misc rc $?"
python profiles/vae_probe.py 128 > gpurun_out/r05b/vae_128_flash.txt 2>&1
LDX_ATTN512=0 python profiles/vae_probe.py 128 > gpurun_out/r05b/vae_128_old.txt 2>&1
python profiles/vae_probe.py 256 > gpurun_out/r05b/vae_256_flash.txt 2>&1
LDX_ATTN512=0 python profiles/vae_probe.py 256 > gpurun_out/r05b/vae_256_old.txt 2>&1
for s in 1 2 4; do LDX_ATTN512_SPLITS=$s python profiles/vae_probe.py 128 2>&1 | grep -E "VAE decode|attn512" > gpurun_out/r05b/vae_128_splits$s.txt; done
grep -E "rel-L2|passed|failed|Error|error" gpurun_out/r05b/t_attn512.log | head -40
tail -3 gpurun_out/r05b/t_misc.log
head -8 gpurun_out/r05b/vae_128_flash.txt gpurun_out/r05b/vae_128_old.txt gpurun_out/r05b/vae_256_flash.txt gpurun_out/r05b/vae_256_old.txt
cat gpurun_out/r05b/vae_128_splits*.txt
