# output stage of a 256-row tile in three parts (stamps 3 -> 5 -> 6 -> 4)
cd /root/repo
B=profiles/ubench/pp_stamp
$B mx 4352 3072 3072 224
$B mx 4352 3072 3072 224 resid
$B bf16 8192 640 640 160
$B bf16 8192 640 640 160 resid
$B bf16 8192 8192 1024 256
$B bf16 32768 320 2880 160 resid
