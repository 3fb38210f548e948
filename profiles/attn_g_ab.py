"""A/B of the generic 32x32x16 attention kernel (LDX_ATTN32G bit mask, read once per process) against the 16x16x32 one."""
import ctypes as C, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ldx_amd as ldx
L = ldx.lib.load()
p = lambda t: C.c_void_p(t.data_ptr())
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)

def run(B, H, N, M, D, dt=torch.bfloat16, reps=10):
    Cc = H * D
    g = torch.Generator(device="cuda").manual_seed(N + D)
    q = torch.randn(B, N, Cc, device="cuda", generator=g).to(dt)
    kv = torch.randn(B, M, 2 * Cc, device="cuda", generator=g).to(dt)
    O = torch.empty(B, N, Cc, device="cuda", dtype=dt)
    code = 0 if dt == torch.bfloat16 else 1
    fn = lambda: L.ldx_op_attention(p(q), Cc, p(kv), 2 * Cc, p(kv[..., Cc:]), 2 * Cc, p(O), Cc, B, H, N, M, D, 1 / math.sqrt(D), 0, code, st())
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    nq = min(N, 300)
    qq = q[:, :nq].float().view(B, nq, H, D).transpose(1, 2)
    k = kv[..., :Cc].float().view(B, M, H, D).transpose(1, 2)
    v = kv[..., Cc:].float().view(B, M, H, D).transpose(1, 2)
    ref = torch.softmax(qq @ k.transpose(-1, -2) / math.sqrt(D), -1) @ v
    got = O[:, :nq].float().view(B, nq, H, D).transpose(1, 2)
    rel = float((got - ref).norm() / ref.norm())
    print(f"32g={os.environ.get('LDX_ATTN32G','0'):>2} B{B} H{H} N{N} M{M} D{D} {str(dt)[6:]}: {ms:.3f} ms {4.0*B*H*N*M*D/ms/1e9:7.1f} TF  rel-L2 {rel:.2e}")

run(2, 8, 4096, 4096, 80)
run(2, 8, 4000, 3001, 80, dt=torch.float16)
run(4, 16, 1024, 1024, 160)
run(1, 24, 4352, 4352, 128)
run(2, 24, 4352, 4352, 128, dt=torch.float16)
run(4, 16, 2048, 777, 64)
