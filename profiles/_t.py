import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kprobe import attn
attn(reps=10)
attn(B=1, H=24, N=4352, D=128, reps=20)
attn(B=2, H=8, N=4096, D=80, reps=20)
attn(B=2, H=8, N=1024, D=160, reps=20)
attn(N=16384, M=77, reps=20)
