timeout 300 python profiles/vae_probe.py 128 2>&1 | grep -E "decode|N3 |N128"
LDX_CONV_PATCH=0 timeout 300 python profiles/vae_probe.py 128 2>&1 | grep -E "decode|N3 |N128"
python - <<'P'
import ctypes as C, math, os, sys, torch
sys.path.insert(0, os.getcwd())
import ldx_amd as ldx
L = ldx.lib.load(); p = lambda t: None if t is None else C.c_void_p(t.data_ptr()); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for (H, W, Cin, Cout) in ((2048, 2048, 64, 3), (1024, 1024, 128, 3), (256, 256, 320, 4)):
    X = torch.randn(1, H, W, Cin, device="cuda").bfloat16(); Wp = (torch.randn(Cout, 9 * Cin, device="cuda") / 30).bfloat16(); Y = torch.zeros(H * W, Cout, device="cuda", dtype=torch.bfloat16)
    run = lambda: L.ldx_op_conv3x3(p(X), Cin, p(Wp), 1, H, W, Cin, Cout, 1, H, W, 0, None, None, 0, None, 0, p(Y), Cout, 0, st)
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize(); print(f"conv {H}x{W} {Cin}->{Cout}: {e0.elapsed_time(e1) * 100:.1f} us")
P
