# per-phase life of a 256-row ping-pong tile (wall-clock stamps inside the product kernel)
cd /root/repo
B=profiles/ubench/pp_stamp
for K in 128 3072 6144 15360; do $B mx 4352 3072 $K 224; done
$B mx 4352 3072 3072 224 resid
$B mx 4352 3072 3072 192
$B mx 4352 3072 3072 128
$B mx 4352 9216 3072 224
$B mx 4352 12288 3072 224
for K in 64 640 1280 5120; do $B bf16 8192 640 $K 160; done
$B bf16 8192 640 640 128
$B bf16 65536 640 640 160
$B bf16 2048 1280 1280 160
$B bf16 32768 320 2880 160
$B bf16 8192 8192 8192 256
