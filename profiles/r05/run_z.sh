timeout 600 python -m pytest tests/test_step_cache_gpu.py tests/test_ops_gpu.py -m gpu -x -q -k "conv3x3 or cfg_denoisers or context_cache" 2>&1 | tail -2
python - <<'P'
import ctypes as C, math, os, sys, torch
sys.path.insert(0, os.getcwd())
import ldx_amd as ldx
L = ldx.lib.load(); p = lambda t: None if t is None else C.c_void_p(t.data_ptr()); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for (B, H, W, Cin, Cout) in ((2, 128, 128, 320, 4), (2, 64, 64, 320, 4), (2, 256, 256, 320, 4)):
    X = torch.randn(B, H, W, Cin, device="cuda").bfloat16(); Wp = (torch.randn(Cout, 9 * Cin, device="cuda") / 30).bfloat16(); Y = torch.zeros(B * H * W, Cout, device="cuda", dtype=torch.float32)
    bias = torch.randn(Cout, device="cuda")
    g = torch.cuda.CUDAGraph()
    run = lambda: L.ldx_op_conv3x3(p(X), Cin, p(Wp), B, H, W, Cin, Cout, 1, H, W, 0, p(bias), None, 0, None, 0, p(Y.bfloat16()), Cout, 0, st)
    Yb = torch.zeros(B * H * W, Cout, device="cuda", dtype=torch.bfloat16)
    run = lambda: L.ldx_op_conv3x3(p(X), Cin, p(Wp), B, H, W, Cin, Cout, 1, H, W, 0, p(bias), None, 0, None, 0, p(Yb), Cout, 0, st)
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize(); us = e0.elapsed_time(e1) * 50
    ref = torch.nn.functional.conv2d(X.float().permute(0, 3, 1, 2), Wp.float().reshape(Cout, 3, 3, Cin).permute(0, 3, 1, 2), bias, padding=1).permute(0, 2, 3, 1).reshape(-1, Cout)
    print(f"conv B{B} {H}x{W} {Cin}->{Cout}: {us:.1f} us  rel {float((Yb.float() - ref).norm() / ref.norm()):.2e}")
P
