"""LayerNorm micro-probe through the C ABI (rows x C as in the SD1.5 levels)."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ldx_amd as ldx
L = ldx.lib.load()
p = lambda t: C.c_void_p(t.data_ptr())
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
for rows, Cn in ((32768, 320), (8192, 640), (2048, 1280), (4352, 3072)):
    X = torch.randn(rows, Cn, device="cuda").bfloat16(); Y = torch.empty_like(X)
    g = torch.ones(Cn, device="cuda"); b = torch.zeros(Cn, device="cuda")
    fn = lambda: L.ldx_op_layernorm(p(X), Cn, p(Y), Cn, rows, Cn, 1e-5, p(g), p(b), 0, st())
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    print(f"LDX_LN_LPR={os.environ.get('LDX_LN_LPR', '-')} ln {rows}x{Cn}: {ms*1e3:.1f} us  {4.0*rows*Cn/ms/1e6:.0f} GB/s")
