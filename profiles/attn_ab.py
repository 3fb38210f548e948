"""A/B of the attention kernel variants (LDX_ATTN_PIPE read once per process -> one process per variant).
Usage: LDX_ATTN_PIPE=<0|22|32|41|42> (only with profiles/experiments/attention_pipelined.hip.txt built in) python profiles/attn_ab.py"""
import ctypes as C, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ldx_amd as ldx
L = ldx.lib.load()
p = lambda t: C.c_void_p(t.data_ptr())
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)

def run(B, H, N, M, D, dt=torch.bfloat16, scale_in=1.0, reps=10):
    Cc = H * D
    g = torch.Generator(device="cuda").manual_seed(N + D)
    qkv = (torch.randn(B, N, 3 * Cc, device="cuda", generator=g) * scale_in).to(dt)
    O = torch.empty(B, N, Cc, device="cuda", dtype=dt)
    code = 0 if dt == torch.bfloat16 else 1
    fn = lambda: L.ldx_op_attention(p(qkv), 3 * Cc, p(qkv[..., Cc:]), 3 * Cc, p(qkv[..., 2 * Cc:]), 3 * Cc, p(O), Cc, B, H, N, M, D, 1 / math.sqrt(D), 0, code, st())
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    # reference on a slice of queries (fp32 math on the 16-bit inputs)
    nq = min(N, 512)
    q = qkv[:, :nq, :Cc].float().view(B, nq, H, D).transpose(1, 2)
    k = qkv[:, :M, Cc:2 * Cc].float().view(B, M, H, D).transpose(1, 2)
    v = qkv[:, :M, 2 * Cc:].float().view(B, M, H, D).transpose(1, 2)
    ref = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(D), -1) @ v
    got = O[:, :nq].float().view(B, nq, H, D).transpose(1, 2)
    rel = float((got - ref).norm() / ref.norm())
    print(f"pipe={os.environ.get('LDX_ATTN_PIPE','0'):>2} B{B} H{H} N{N} M{M} D{D} x{scale_in}: {ms:.3f} ms {4.0*B*H*N*M*D/ms/1e9:7.1f} TF  rel-L2 {rel:.2e}")

run(2, 8, 16384, 16384, 40)
run(2, 8, 16384, 16384, 40, scale_in=4.0)     # peaky softmax: exercises the lazy-max rescale branch
run(2, 8, 4096, 4096, 40, dt=torch.float16)
run(1, 8, 3000, 3000, 40)                      # ragged q and kv tails
