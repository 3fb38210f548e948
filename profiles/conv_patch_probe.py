"""Time ldx_op_conv3x3 on the shapes conv_patch.hip takes (ESRGAN dense-block convs on a 512^2 tile, the 2048^2 tail, the VAE's 128-channel level).
Run twice — default and LDX_CONV_PATCH=0 (the implicit-GEMM tiles) — to compare; prints us per launch and TFLOP/s."""
import ctypes as C, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ldx_amd as ldx
L = ldx.lib.load()
p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
SHAPES = [(512, 512, 64, 32, 192), (512, 512, 128, 32, 192), (512, 512, 192, 32, 192), (512, 512, 192, 64, 192), (512, 512, 64, 64, 64),
          (2048, 2048, 64, 64, 64), (1024, 1024, 128, 128, 128), (1024, 1024, 256, 128, 256), (2048, 2048, 128, 128, 128),
          (256, 512, 64, 32, 192), (1024, 512, 64, 32, 192), (2048, 512, 64, 32, 192), (256, 512, 192, 32, 192), (1024, 512, 192, 32, 192)]      # 9-13: tiles per workgroup 1 / 4 / 8: fixed cost per launch vs cost per tile
only = int(sys.argv[1]) if len(sys.argv) > 1 else -1
lo = int(sys.argv[2]) if len(sys.argv) > 2 else -1
for idx, (H, W, Cin, Cout, ld) in enumerate(SHAPES):
    if only >= 0 and idx != only and not (lo >= 0 and only <= idx <= lo): continue
    X = torch.randn(1, H, W, ld, device="cuda").bfloat16()
    Wp = (torch.randn(Cout, 9 * Cin, device="cuda") / math.sqrt(9 * Cin)).bfloat16()
    bias = torch.randn(Cout, device="cuda")
    Y = torch.zeros(H * W, Cout, device="cuda", dtype=torch.bfloat16)
    run = lambda: ldx.lib.check(L.ldx_op_conv3x3(p(X), ld, p(Wp), 1, H, W, Cin, Cout, 1, H, W, 0, p(bias), None, 0, None, 0, p(Y), Cout, 0, st), "conv")
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    fl = 2.0 * H * W * 9 * Cin * Cout
    print(f"conv {H}x{W} Cin {Cin:3d} (ld {ld:3d}) -> {Cout:3d}: {us:8.1f} us  {fl / us / 1e6:6.0f} TFLOP/s")
