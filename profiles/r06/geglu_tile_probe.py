"""GEGLU up-projections of the 64^2 / 32^2 levels: 256 x 128 (round 2-5) against 256 x 256 ping-pong tiles (round 6).  LDX_GEMM_TILE is read once per process.
Usage: [LDX_GEMM_TILE=256128] python profiles/r06/geglu_tile_probe.py"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ldx_amd as ldx
L = ldx.lib.load(); p = lambda t: None if t is None else C.c_void_p(t.data_ptr()); st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
for (M, N, K) in ((8192, 5120, 640), (2048, 10240, 1280), (512, 10240, 1280), (65536, 5120, 640), (16384, 10240, 1280)):
    A = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") / 30).bfloat16(); bias = torch.randn(N, device="cuda")
    Cc = torch.empty(M, N // 2, device="cuda", dtype=torch.bfloat16)
    run = lambda: L.ldx_op_gemm(p(A), K, p(W), M, N, K, p(bias), None, 0, 1, 1, None, 0, p(Cc), N // 2, None, 0, 0, st())
    for _ in range(3): run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20): run()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize(); us = e0.elapsed_time(e1) * 10
    print(f"tile {os.environ.get('LDX_GEMM_TILE', 'auto'):>7} geglu M{M} N{N} K{K}: {us:7.1f} us  {2.0 * M * N * K / us / 1e6:6.0f} TF", flush=True)
