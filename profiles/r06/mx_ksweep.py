"""MX fp8 256-row ping-pong GEMM: fixed cost per tile vs cost per K-tile on the Flux-dev token count (M = 4352), through the C ABI.
Weights rotate through enough copies (> 600 MB) that no launch finds its operands in L2 / MALL, as inside the forward.
Variants: plain 16-bit output; 16-bit output + residual read (the gated-residual shape of proj / mlp2 / linear2); quantised output (C8).
Usage: python profiles/r06/mx_ksweep.py [N [K,K,...]]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ldx_amd as ldx  # noqa: F401
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kprobe import p, st, L  # noqa: E402

M = 4352
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3072


def graph_time(fns, reps=3):
    for f in fns[:2]:
        f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for f in fns:
            f()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * len(fns))


KS = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [512, 1024, 2048, 3072, 6144, 12288, 15360]
for K in KS:
    ncopy = max(4, int(6e8 // ((M + N) * K)) + 1)
    A8 = [torch.randint(0, 120, (M, K), device="cuda", dtype=torch.uint8) for _ in range(ncopy)]
    W8 = [torch.randint(0, 120, (N, K), device="cuda", dtype=torch.uint8) for _ in range(ncopy)]
    SA = torch.full((K // 128, M), 0x7f7f7f7f, device="cuda", dtype=torch.int32)
    SW = torch.full((K // 128, N), 0x7a7a7a7a, device="cuda", dtype=torch.int32)
    Cc = [torch.zeros(M, N, device="cuda", dtype=torch.bfloat16) for _ in range(min(ncopy, 8))]
    C8 = torch.empty(M, N, device="cuda", dtype=torch.uint8)
    SC = torch.zeros(N // 128, M, device="cuda", dtype=torch.int32)
    bias = torch.zeros(N, device="cuda")
    res = {}
    for name in ("plain", "resid", "quant"):
        fns = []
        for i in range(ncopy * 2):
            a, w, c = A8[i % ncopy], W8[i % ncopy], Cc[i % len(Cc)]
            if name == "plain":
                fns.append(lambda a=a, w=w, c=c: L.ldx_op_gemm_mx(p(a), K, p(SA), M, p(w), p(SW), N, M, N, K, p(bias), 0, None, 0, p(c), N, None, 0, None, 0, None, 0, 0, st()))
            elif name == "resid":
                fns.append(lambda a=a, w=w, c=c: L.ldx_op_gemm_mx(p(a), K, p(SA), M, p(w), p(SW), N, M, N, K, p(bias), 0, p(c), N, p(c), N, None, 0, None, 0, None, 0, 0, st()))
            else:
                fns.append(lambda a=a, w=w: L.ldx_op_gemm_mx(p(a), K, p(SA), M, p(w), p(SW), N, M, N, K, p(bias), 2, None, 0, None, 0, None, 0, p(C8), N, p(SC), M, 0, st()))
        assert fns[0]() == 0
        res[name] = graph_time(fns) * 1e3
    fl = 2.0 * M * N * K
    print(f"M{M} N{N} K{K:6d} ({K // 128:3d} K-tiles, {ncopy} operand sets): " + "  ".join(f"{k} {v:7.1f} us {fl / v / 1e6:6.0f} TF" for k, v in res.items()), flush=True)
    del A8, W8, Cc
