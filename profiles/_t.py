import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kprobe import gemm, conv
gemm(8192, 8192, 8192, 5)
gemm(4096, 12288, 3072, 20)
gemm(32768, 320, 320, 50)
gemm(8192, 640, 640, 50)
gemm(2048, 1280, 5120, 50)
conv(2, 128, 320, 320, 20)
conv(2, 64, 640, 640, 20)
conv(2, 32, 1280, 1280, 20)
