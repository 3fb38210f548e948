# round 6, first GPU call: today's baseline (per-shape table bs=1 and CFG batch 16) + planner switches that already exist
mkdir -p gpurun_out/r06a
python profiles/shape_probe.py 128 bf16 2 > gpurun_out/r06a/shape_b2.txt 2>&1; head -3 gpurun_out/r06a/shape_b2.txt
python profiles/shape_probe.py 128 bf16 16 > gpurun_out/r06a/shape_b16_default.txt 2>&1; head -3 gpurun_out/r06a/shape_b16_default.txt
LDX_ROWBLOCK_MINWG=1000000 python profiles/shape_probe.py 128 bf16 16 > gpurun_out/r06a/shape_b16_norowblock.txt 2>&1; head -3 gpurun_out/r06a/shape_b16_norowblock.txt
LDX_ROWBLOCK_MINWG=1000000 LDX_LNFOLD_MAXROWS=100000000 python profiles/shape_probe.py 128 bf16 16 > gpurun_out/r06a/shape_b16_norowblock_fold.txt 2>&1; head -3 gpurun_out/r06a/shape_b16_norowblock_fold.txt
LDX_PP=2 python profiles/shape_probe.py 128 bf16 16 > gpurun_out/r06a/shape_b16_pp2.txt 2>&1; head -3 gpurun_out/r06a/shape_b16_pp2.txt
