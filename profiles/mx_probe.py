"""MX fp8 vs bf16 GEMM on the Flux-dev shapes, and the stand-alone quantiser, through the C ABI.
Usage: python profiles/mx_probe.py [reps]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ldx_amd as ldx
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kprobe import timeit, p, st, L  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
SHAPES = [(4096, 9216, 3072), (4096, 3072, 3072), (4096, 12288, 3072), (4096, 3072, 12288), (4352, 9216, 3072), (4352, 12288, 3072),
          (4352, 3072, 15360), (256, 9216, 3072), (256, 3072, 12288), (8192, 8192, 8192)]
for M, N, K in SHAPES:
    A = torch.randn(M, K, device="cuda").bfloat16()
    W = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
    Cc = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    A8 = torch.empty(M, K, device="cuda", dtype=torch.uint8)
    W8 = torch.empty(N, K, device="cuda", dtype=torch.uint8)
    SA = torch.zeros(K // 128, M, device="cuda", dtype=torch.int32)
    SW = torch.zeros(K // 128, N, device="cuda", dtype=torch.int32)
    q = lambda: L.ldx_op_mx_quant(p(A), K, M, K, p(A8), K, p(SA), M, 0, st())
    assert q() == 0 and L.ldx_op_mx_quant(p(W), K, N, K, p(W8), K, p(SW), N, 0, st()) == 0
    f16 = lambda: L.ldx_op_gemm(p(A), K, p(W), M, N, K, None, None, 0, 1, 0, None, 0, p(Cc), N, None, 0, 0, st())
    f8 = lambda: L.ldx_op_gemm_mx(p(A8), K, p(SA), M, p(W8), p(SW), N, M, N, K, None, 0, None, 0, p(Cc), N, None, 0, None, 0, None, 0, 0, st())
    assert f16() == 0
    ref = Cc.float().clone()
    assert f8() == 0
    err = float((Cc.float() - ref).norm() / ref.norm())
    t16, t8, tq = timeit(f16, reps), timeit(f8, reps), timeit(q, reps)
    fl = 2.0 * M * N * K
    print(f"{M:5d}x{N:5d}x{K:5d}: bf16 {t16 * 1e3:8.1f} us {fl / t16 / 1e9:7.1f} TF | mx {t8 * 1e3:8.1f} us {fl / t8 / 1e9:7.1f} TF  x{t16 / t8:.2f} | "
          f"quant A {tq * 1e3:6.1f} us ({M * K * 3 / tq / 1e6:.0f} GB/s) | mx-vs-bf16 rel {err:.3e}", flush=True)
