"""Micro-probe: time single ops through the C ABI (HIP events on torch's current stream, which is the launch
stream).  Usage: python profiles/kprobe.py attn|gemm|conv|gn [reps]"""
import ctypes as C
import math
import sys

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ldx_amd as ldx

L = ldx.lib.load()
p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def timeit_graph(fn, reps=50, warm=3):
    """Host launch cost (ctypes from Python: >= 10 us per call) out of the picture: capture `reps` launches in a hipGraph, replay."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * reps)


def attn(B=2, H=8, N=16384, M=None, D=40, reps=10):
    M = M or N
    C_ = H * D
    qkv = torch.randn(B, N, 3 * C_, device="cuda").bfloat16()
    O = torch.empty(B, N, C_, device="cuda", dtype=torch.bfloat16)
    fn = lambda: L.ldx_op_attention(p(qkv), 3 * C_, p(qkv[..., C_:]), 3 * C_, p(qkv[..., 2 * C_:]), 3 * C_, p(O), C_, B, H, N, M, D, 1 / math.sqrt(D), 0, 0, st())
    ms = timeit(fn, reps)
    fl = 4.0 * B * H * N * M * D
    print(f"attn B{B} H{H} N{N} M{M} D{D}: {ms:.3f} ms  {fl / ms / 1e9:.1f} TFLOP/s")


def gemm(M=32768, N=320, K=320, reps=20, geglu=0):
    A = torch.randn(M, K, device="cuda").bfloat16()
    W = torch.randn(N, K, device="cuda").bfloat16()
    No = N // 2 if geglu else N                      # GEGLU: value / gate column pairs -> N / 2 outputs
    Cc = torch.empty(M, No, device="cuda", dtype=torch.bfloat16)
    fn = lambda: L.ldx_op_gemm(p(A), K, p(W), M, N, K, None, None, 0, 1, geglu, None, 0, p(Cc), No, None, 0, 0, st())
    assert fn() == 0
    ms = timeit(fn, reps)
    print(f"gemm {M}x{N}x{K}{' geglu' if geglu else ''}: {ms * 1000:.1f} us  {2.0 * M * N * K / ms / 1e9:.1f} TFLOP/s")


def conv(B=2, H=128, Cin=320, Cout=320, reps=10):
    X = torch.randn(B, H, H, Cin, device="cuda").bfloat16()
    W = torch.randn(Cout, 9 * Cin, device="cuda").bfloat16()
    Y = torch.empty(B * H * H, Cout, device="cuda", dtype=torch.bfloat16)
    fn = lambda: L.ldx_op_conv3x3(p(X), Cin, p(W), B, H, H, Cin, Cout, 1, H, H, 0, None, None, 0, None, 0, p(Y), Cout, 0, st())
    ms = timeit(fn, reps)
    print(f"conv B{B} {H}x{H} {Cin}->{Cout}: {ms * 1000:.1f} us  {2.0 * B * H * H * Cout * 9 * Cin / ms / 1e9:.1f} TFLOP/s")


def gn(B=2, HW=16384, Cn=320, reps=20):
    X = torch.randn(B, HW, Cn, device="cuda").bfloat16()
    Y = torch.empty_like(X)
    g = torch.ones(Cn, device="cuda"); b = torch.zeros(Cn, device="cuda")
    ws = torch.zeros(L.ldx_op_groupnorm_workspace_floats(B, 32), device="cuda")
    fn = lambda: L.ldx_op_groupnorm(p(X), Cn, p(Y), Cn, B, HW, Cn, 32, 1e-5, 1, p(g), p(b), p(ws), 0, st())
    ms = timeit(fn, reps)
    print(f"gn B{B} HW{HW} C{Cn}: {ms * 1000:.1f} us  {3 * 2.0 * B * HW * Cn / ms / 1e6:.1f} GB/s (3 passes)")


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what == "conv1":
        conv(reps=3)
    if what == "gemm1":
        gemm(8192, 8192, 8192, reps=2)
    if what == "attn1":
        attn(reps=3)
    if what == "attnsmall":    # the short-sequence launches of the deeper SD1.5 levels (grids below one workgroup per CU)
        for (n, d) in ((1024, 160), (1024, 80), (256, 160), (4096, 80)):
            qkv = torch.randn(2, n, 3 * 8 * d, device="cuda").bfloat16(); Cn = 8 * d
            O = torch.empty(2, n, Cn, device="cuda", dtype=torch.bfloat16)
            fn = lambda: L.ldx_op_attention(p(qkv), 3 * Cn, p(qkv[..., Cn:]), 3 * Cn, p(qkv[..., 2 * Cn:]), 3 * Cn, p(O), Cn, 2, 8, n, n, d, 1 / math.sqrt(d), 0, 0, st())
            us = timeit_graph(fn, 40) * 1e3
            print(f"attn B2 H8 N{n} D{d}: {us:.1f} us  {4.0 * 2 * 8 * n * n * d / us / 1e6:.0f} TFLOP/s")
    if what == "attn128":      # one Flux joint-attention launch
        attn(B=1, H=24, N=4352, D=128, reps=3)
    if what == "gemm640":      # a plain mid-size projection (SD1.5 64^2 level)
        gemm(8192, 640, 640, reps=3)
    if what == "deep1":        # a weight-streaming split-K conv of the 16^2 level (M = 512)
        conv(H=16, Cin=1280, Cout=1280, reps=3)
    if what in ("attn", "all"):
        attn(); attn(N=4096, D=80); attn(N=1024, D=160); attn(N=16384, M=77)
    if what in ("gemm", "all"):
        for s in ((32768, 320, 320), (32768, 960, 320), (32768, 2560, 320), (32768, 320, 1280), (8192, 640, 640), (8192, 5120, 640), (8192, 640, 2560),
                  (2048, 1280, 1280), (2048, 10240, 1280), (2048, 1280, 5120), (8192, 8192, 8192)):
            gemm(*s)
    if what in ("conv", "all"):
        conv(); conv(H=64, Cin=640, Cout=640); conv(H=32, Cin=1280, Cout=1280); conv(H=16, Cin=1280, Cout=1280); conv(H=16, Cin=2560, Cout=1280)
    if what in ("gn", "all"):
        gn(); gn(HW=4096, Cn=640); gn(HW=1024, Cn=1280); gn(HW=256, Cn=2560)
    if what == "geglu":    # the three GEGLU projections of an SD1.5 step (run with LDX_GEMM_TILE unset / =256128)
        gemm(32768, 2560, 320, geglu=1); gemm(8192, 5120, 640, geglu=1); gemm(2048, 10240, 1280, geglu=1)
    if what == "ksweep":
        for (M, N) in ((32768, 320), (8192, 640), (2048, 1280)):
            for K in (64, 128, 320, 640, 1280, 2560):
                gemm(M, N, K)
    if what == "pp":       # shapes for the 256-row ping-pong tiles (run with LDX_GEMM_TILE unset / =256128 / 256160 / 256256)
        for s in ((32768, 320, 320), (32768, 960, 320), (32768, 320, 1280), (32768, 2560, 320), (8192, 640, 640), (8192, 640, 2560),
                  (4096, 3072, 3072), (4352, 9216, 3072), (4352, 3072, 15360), (4096, 12288, 3072), (4096, 4096, 4096), (8192, 8192, 8192)):
            gemm(*s)
        conv(); conv(Cin=640, Cout=320); conv(Cin=960, Cout=320); conv(H=64, Cin=640, Cout=640); conv(H=64, Cin=1280, Cout=640)
        conv(B=1, H=512, Cin=256, Cout=256); conv(B=1, H=1024, Cin=128, Cout=128)
    if what == "cold":     # hot vs cold operands: rotate over enough distinct buffers to exceed the 256 MiB Infinity Cache
        def gemm_cold(M, N, K, nbuf_a, nbuf_w, reps=200):
            As = [torch.randn(M, K, device="cuda").bfloat16() for _ in range(nbuf_a)]
            Ws = [torch.randn(N, K, device="cuda").bfloat16() for _ in range(nbuf_w)]
            Cc = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            i = [0]
            def fn():
                a = As[i[0] % nbuf_a]; w = Ws[i[0] % nbuf_w]; i[0] += 1
                L.ldx_op_gemm(p(a), K, p(w), M, N, K, None, None, 0, 1, 0, None, 0, p(Cc), N, None, 0, 0, st())
            ms = timeit_graph(fn, reps)
            print(f"gemm {M}x{N}x{K} A x{nbuf_a} W x{nbuf_w}: {ms * 1000:.1f} us  {2.0 * M * N * K / ms / 1e9:.1f} TFLOP/s")
        for (M, N, K) in ((2048, 1280, 1280), (8192, 640, 640), (32768, 320, 320), (512, 1280, 1280)):
            wn = max(1, int(400e6 / (N * K * 2))); an = max(1, int(400e6 / (M * K * 2)))
            gemm_cold(M, N, K, 1, 1); gemm_cold(M, N, K, 1, wn); gemm_cold(M, N, K, an, 1); gemm_cold(M, N, K, an, wn)
    if what == "rowblock":   # the C = 320 row-block kernels (rowgemm LN + q|k|v, xattn_block, ff_block) at M = 32768, hipGraph-timed
        M, Cc, inner, Mk = 32768, 320, 1280, 77
        rn = lambda *s_: torch.randn(*s_, device="cuda")
        h = rn(M, Cc).bfloat16(); gamma, beta = 1 + 0.1 * rn(Cc), 0.1 * rn(Cc)
        Wq = (rn(Cc, Cc) / math.sqrt(Cc)).bfloat16(); Wo = (rn(Cc, Cc) / math.sqrt(Cc)).bfloat16(); bo = 0.1 * rn(Cc)
        Wqkv = (rn(3 * Cc, Cc) / math.sqrt(Cc)).bfloat16(); qkv = torch.empty(M, 3 * Cc, device="cuda", dtype=torch.bfloat16)
        kv = rn(2 * Mk, 2 * Cc).bfloat16(); h2 = rn(M, Cc).bfloat16()
        W1 = (rn(2 * inner, Cc) / math.sqrt(Cc)).bfloat16(); b1 = 0.1 * rn(2 * inner); W2 = (rn(Cc, inner) / math.sqrt(inner)).bfloat16(); b2 = 0.1 * rn(Cc)
        for name, fl, fn in (
            ("rowgemm LN + q|k|v (N = 960)", 2.0 * M * 960 * Cc, lambda: L.ldx_op_rowgemm(p(h), Cc, p(qkv), 3 * Cc, M, 3 * Cc, Cc, p(Wqkv), None, None, 0, 1, p(gamma), p(beta), 1e-5, None, 0, 0, 0, st())),
            ("rowgemm to_out + residual (N = 320)", 2.0 * M * Cc * Cc, lambda: L.ldx_op_rowgemm(p(h), Cc, p(h2), Cc, M, Cc, Cc, p(Wq), p(bo), p(h2), Cc, 0, None, None, 1e-5, None, 0, 0, 0, st())),
            ("xattn_block (77 keys)", 4.0 * M * Cc * Cc + 4.0 * M * Mk * Cc, lambda: L.ldx_op_xattn_block(p(h), Cc, M, M // 2, Cc, 8, p(gamma), p(beta), 1e-5, p(Wq), p(Wo), p(bo), p(kv), 2 * Cc, p(kv[:, Cc:]), 2 * Cc, Mk, 1 / math.sqrt(40), 0, st())),
            ("ff_block (inner 1280)", 2.0 * M * Cc * 3 * inner, lambda: L.ldx_op_ff_block(p(h), Cc, M, Cc, inner, p(gamma), p(beta), 1e-5, p(W1), p(b1), p(W2), p(b2), 0, st()))):
            ms = timeit_graph(fn, 20)
            print(f"{name}: {ms * 1000:.1f} us  {fl / ms / 1e9:.1f} TFLOP/s")
    if what == "deep":     # weight-streaming convs / GEMMs of the deep UNet levels (M = 512 / 2048): run with LDX_GEMM_TILE / LDX_SPLITK sweeps
        def convg(B, H, Cin, Cout):
            X = torch.randn(B, H, H, Cin, device="cuda").bfloat16()
            Ws = [torch.randn(Cout, 9 * Cin, device="cuda").bfloat16() for _ in range(max(1, int(600e6 / (Cout * 9 * Cin * 2))))]   # cold weights
            Y = torch.empty(B * H * H, Cout, device="cuda", dtype=torch.bfloat16)
            i = [0]
            def fn():
                w = Ws[i[0] % len(Ws)]; i[0] += 1
                L.ldx_op_conv3x3(p(X), Cin, p(w), B, H, H, Cin, Cout, 1, H, H, 0, None, None, 0, None, 0, p(Y), Cout, 0, st())
            ms = timeit_graph(fn, reps=len(Ws) * 2)
            print(f"conv B{B} {H}x{H} {Cin}->{Cout} (cold W x{len(Ws)}): {ms * 1000:.1f} us  {2.0 * B * H * H * Cout * 9 * Cin / ms / 1e9:.1f} TFLOP/s  W {Cout * 9 * Cin * 2 / ms / 1e6:.0f} GB/s")
        convg(2, 16, 1280, 1280); convg(2, 16, 2560, 1280); convg(2, 32, 1280, 1280); convg(2, 32, 2560, 1280); convg(2, 32, 1920, 1280); convg(2, 64, 1280, 640)
    if what == "shortk":   # the N = K = C projections of the transformer blocks, hipGraph-timed (run under LDX_GEMM_TILE=... to compare tiles)
        for (M, N, K) in ((32768, 320, 320), (8192, 640, 640), (2048, 1280, 1280), (32768, 960, 320), (8192, 1920, 640), (32768, 320, 1280), (8192, 640, 2560), (2048, 1280, 5120),
                          (2048, 3840, 1280), (512, 1280, 1280), (16384, 512, 512), (4352, 3072, 3072)):
            A = torch.randn(M, K, device="cuda").bfloat16(); W = torch.randn(N, K, device="cuda").bfloat16()
            R = torch.randn(M, N, device="cuda").bfloat16(); Cc = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            fn = lambda: L.ldx_op_gemm(p(A), K, p(W), M, N, K, None, None, 0, 1, 0, p(R), N, p(Cc), N, None, 0, 0, st())
            ms = timeit_graph(fn, 40)
            ref = A.float() @ W.float().t() + R.float()
            fn(); torch.cuda.synchronize()
            err = float((Cc.float() - ref).norm() / ref.norm())
            print(f"gemm+residual {M}x{N}x{K}: rel-L2 {err:.1e}  {ms * 1000:.1f} us  {2.0 * M * N * K / ms / 1e9:.1f} TFLOP/s  {(M * K + 2 * M * N) * 2 / ms / 1e6:.0f} GB/s")
    if what == "small":    # the latency-bound projections of the 512^2 / 1024^2 steps (64 x 64 tiles): run with LDX_RING64=0 / 1
        for s in ((512, 1280, 1280), (2048, 640, 640), (8192, 320, 320), (8192, 320, 1280), (2048, 1280, 1280), (512, 1280, 5120), (128, 1280, 1280), (8192, 640, 640), (8192, 640, 2560)):
            ms = timeit_graph(lambda: None, 1) if False else None
            M, N, K = s
            A = torch.randn(M, K, device="cuda").bfloat16(); Wt = (torch.randn(N, K, device="cuda") / math.sqrt(K)).bfloat16(); Cc = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            bias = torch.randn(N, device="cuda"); R = torch.randn(M, N, device="cuda").bfloat16()
            noepi = os.environ.get("NOEPI") == "1"      # no bias / residual: what the epilogue operands cost
            fn = lambda: L.ldx_op_gemm(p(A), K, p(Wt), M, N, K, None if noepi else p(bias), None, 0, 1, 0, None if noepi else p(R), N, p(Cc), N, None, 0, 0, st())
            ms = timeit_graph(fn, 50)
            print(f"gemm {M}x{N}x{K} + bias + residual: {ms * 1000:.1f} us  {2.0 * M * N * K / ms / 1e9:.1f} TFLOP/s")
    if what == "convepi":  # what the epilogue operands (bias, per-batch row vector, residual) cost on the level-0 conv shapes
        for (B, H, Cin, Cout) in ((2, 128, 320, 320), (2, 128, 640, 320), (2, 64, 640, 640), (2, 32, 1280, 1280)):
            X = torch.randn(B, H, H, Cin, device="cuda").bfloat16(); W = (torch.randn(Cout, 9 * Cin, device="cuda") / math.sqrt(9 * Cin)).bfloat16()
            Y = torch.empty(B * H * H, Cout, device="cuda", dtype=torch.bfloat16); bias = torch.randn(Cout, device="cuda"); rvec = torch.randn(B, Cout, device="cuda")
            R = torch.randn(B * H * H, Cout, device="cuda").bfloat16()
            for name, b_, rv_, r_ in (("plain", None, None, None), ("bias", bias, None, None), ("bias+rowvec", bias, rvec, None), ("bias+residual", bias, None, R), ("all", bias, rvec, R)):
                fn = lambda: L.ldx_op_conv3x3(p(X), Cin, p(W), B, H, H, Cin, Cout, 1, H, H, 0, p(b_), p(rv_), Cout, p(r_), Cout, p(Y), Cout, 0, st())
                ms = timeit_graph(fn, 20)
                print(f"conv B{B} {H}x{H} {Cin}->{Cout} {name:14s}: {ms * 1000:.1f} us  {2.0 * B * H * H * Cout * 9 * Cin / ms / 1e9:.1f} TFLOP/s")
    if what == "convk":    # fixed cost of a level-0 conv launch: time against Cin at fixed M = 32768, N = 320 (graph-timed)
        for Cin in (64, 128, 192, 320, 640, 960):
            B, H, Cout = 2, 128, 320
            X = torch.randn(B, H, H, Cin, device="cuda").bfloat16(); W = (torch.randn(Cout, 9 * Cin, device="cuda") / math.sqrt(9 * Cin)).bfloat16()
            Y = torch.empty(B * H * H, Cout, device="cuda", dtype=torch.bfloat16)
            fn = lambda: L.ldx_op_conv3x3(p(X), Cin, p(W), B, H, H, Cin, Cout, 1, H, H, 0, None, None, 0, None, 0, p(Y), Cout, 0, st())
            ms = timeit_graph(fn, 20)
            print(f"conv B{B} {H}x{H} {Cin}->{Cout} K {9 * Cin}: {ms * 1000:.1f} us  {2.0 * B * H * H * Cout * 9 * Cin / ms / 1e9:.1f} TFLOP/s")
