"""Time the MX fp8 attention pieces at the Flux shape [1, 24, 4352, 128] against the 16-bit D = 128 kernel.  Usage: python profiles/attn_mx_probe.py [N=4352]"""
import ctypes as C
import math
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ldx_amd as ldx

L = ldx.lib.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4352
H = 24
p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def timeit(fn, reps=20):
    for _ in range(3):
        assert fn() == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


qkv = torch.randn(N, 3 * H * 128, device="cuda").bfloat16()
Cc = H * 128
O = torch.empty(N, Cc, device="cuda", dtype=torch.bfloat16)
scale = 1 / math.sqrt(128)
t16 = timeit(lambda: L.ldx_op_attention(p(qkv), 3 * Cc, p(qkv[:, Cc:]), 3 * Cc, p(qkv[:, 2 * Cc:]), 3 * Cc, p(O), Cc, 1, H, N, N, 128, scale, 0, 0, st))
q8 = torch.empty(N, Cc, device="cuda", dtype=torch.uint8); k8 = torch.empty_like(q8)
sq = torch.empty(H, N, device="cuda", dtype=torch.int32); sk = torch.empty_like(sq)
Lp = (N + 127) // 128 * 128
v8t = torch.empty(1, H, 128, Lp, device="cuda", dtype=torch.uint8); sv = torch.empty(1, H, Lp // 128, 128, device="cuda", dtype=torch.int32)
qs = torch.ones(128, device="cuda"); cosT = torch.rand(N, 64, device="cuda"); sinT = torch.rand(N, 64, device="cuda")
t_rope = timeit(lambda: L.ldx_op_qk_norm_rope_mx(p(qkv), 3 * Cc, N, N, H, p(qs), p(qs), p(cosT), p(sinT), 1e-6, p(q8), p(k8), Cc, p(sq), p(sk), N, 0, st))
t_vt = timeit(lambda: L.ldx_op_mx_vt_quant(p(qkv[:, 2 * Cc:]), 3 * Cc, 1, H, N, p(v8t), p(sv), Lp, 0, st))
t8 = timeit(lambda: L.ldx_op_attention_fp8(p(q8), Cc, p(sq), N, p(k8), Cc, p(sk), N, p(v8t), p(sv), Lp, p(O), Cc, None, 0, None, 0, 1, H, N, N, scale, 0, st))
fl = 4.0 * H * N * N * 128
print(f"N {N} H {H}: 16-bit attention {t16:.1f} us ({fl / t16 / 1e6:.0f} TF) | MX fp8 attention {t8:.1f} us ({fl / t8 / 1e6:.0f} TF) + rope_mx {t_rope:.1f} us + V transposing quantiser {t_vt:.1f} us")
