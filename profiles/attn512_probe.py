"""Time the D = 512 flash attention (csrc/attn512.hip) through the C ABI.  Usage: python profiles/attn512_probe.py [N=16384] [reps=10]"""
import ctypes as C
import math
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ldx_amd as ldx

L = ldx.lib.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
p = lambda t: C.c_void_p(t.data_ptr())
qkv = torch.randn(1, N, 1536, device="cuda").bfloat16()
O = torch.empty(1, N, 512, device="cuda", dtype=torch.bfloat16)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
fn = lambda: L.ldx_op_attention(p(qkv), 1536, p(qkv[..., 512:]), 1536, p(qkv[..., 1024:]), 1536, p(O), 512, 1, 1, N, N, 512, 1 / math.sqrt(512), 0, 0, st)
for _ in range(3):
    assert fn() == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    fn()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print(f"attn512 N{N} ABL={os.environ.get('LDX_ATTN512_ABL', '0')} SPLITS={os.environ.get('LDX_ATTN512_SPLITS', 'auto')}: {ms:.3f} ms  {4.0 * N * N * 512 / ms / 1e9:.0f} TFLOP/s")
