"""Plain GEMMs of the 64^2 / 32^2 levels under forced tiles / split-K (LDX_GEMM_TILE, LDX_SPLITK are read once per process).  hipGraph-timed, residual operand as in the step.
Usage: [LDX_GEMM_TILE=..] [LDX_SPLITK=..] python profiles/r06/gemm_sweep_probe.py"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ldx_amd as ldx
L = ldx.lib.load(); p = lambda t: None if t is None else C.c_void_p(t.data_ptr()); st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
tag = f"tile {os.environ.get('LDX_GEMM_TILE', 'auto'):>7} sk {os.environ.get('LDX_SPLITK', 'auto'):>4}"
for (M, N, K, res) in ((8192, 640, 2560, 1), (2048, 1280, 5120, 1), (2048, 3840, 1280, 0), (2048, 1280, 1280, 1), (8192, 640, 640, 1)):
    A = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") / 30).bfloat16(); bias = torch.randn(N, device="cuda")
    R = torch.randn(M, N, device="cuda").bfloat16() if res else None
    Cc = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    run = lambda: L.ldx_op_gemm(p(A), K, p(W), M, N, K, p(bias), None, 0, 1, 0, p(R), N if res else 0, p(Cc), N, None, 0, 0, st())
    assert run() == 0
    for _ in range(3): run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20): run()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize(); us = e0.elapsed_time(e1) * 10
    print(f"{tag} gemm M{M} N{N} K{K}{' +R' if res else ''}: {us:7.1f} us  {2.0 * M * N * K / us / 1e6:6.0f} TF", flush=True)
