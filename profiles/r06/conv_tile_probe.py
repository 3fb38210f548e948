"""Which tile for the shared prefix's 3x3 convs (M = 16384, N = 320, K = 2880: one image at 128^2)?  LDX_GEMM_TILE is read once per process: run per tile.
Usage: LDX_GEMM_TILE=<BM*1000+BN> python profiles/r06/conv_tile_probe.py"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ldx_amd as ldx
L = ldx.lib.load(); p = lambda t: None if t is None else C.c_void_p(t.data_ptr()); st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
for (B, H, W, Cin, Cout) in ((1, 128, 128, 320, 320), (2, 128, 128, 320, 320), (1, 128, 128, 64, 320)):
    X = torch.randn(B, H, W, Cin, device="cuda").bfloat16(); Wp = (torch.randn(Cout, 9 * Cin, device="cuda") / 30).bfloat16()
    bias = torch.randn(Cout, device="cuda"); Yb = torch.zeros(B * H * W, Cout, device="cuda", dtype=torch.bfloat16)
    run = lambda: L.ldx_op_conv3x3(p(X), Cin, p(Wp), B, H, W, Cin, Cout, 1, H, W, 0, p(bias), None, 0, None, 0, p(Yb), Cout, 0, st())
    for _ in range(3): run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20): run()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize(); us = e0.elapsed_time(e1) * 10
    fl = 2.0 * B * H * W * Cout * 9 * Cin
    print(f"tile {os.environ.get('LDX_GEMM_TILE', 'auto'):>7} conv B{B} {H}x{W} {Cin}->{Cout}: {us:7.1f} us  {fl / us / 1e6:6.0f} TF", flush=True)
