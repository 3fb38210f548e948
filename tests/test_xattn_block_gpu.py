"""The fused cross-attention sub-block (csrc/xattn_block.hip, ldx_op_xattn_block; reference transformer.py:186-245 attn2 +
Attention.py:100-124): LayerNorm + q projection + attention over <= 80 context keys + out projection + bias + residual in one launch,
against (a) fp32 torch on the same 16-bit inputs and (b) the four separate ops it replaces (ldx_op_layernorm, ldx_op_gemm,
ldx_op_attention, ldx_op_gemm with residual), which round at the same places."""
import ctypes as C
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DT = {"bf16": (torch.bfloat16, 0), "f16": (torch.float16, 1)}
_p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
_st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("B,N,Mk", [(2, 1024, 77), (1, 256, 80), (3, 128, 13), (2, 16384, 77)])
def test_xattn_block_vs_torch_and_separate_ops(ldx, ldx_lib, dt, B, N, Mk):
    L = ldx_lib
    td, code = DT[dt]
    Cc, Hh, D = 320, 8, 40
    M = B * N
    g = torch.Generator(device="cuda").manual_seed(B * 1000 + N + Mk)
    rn = lambda *s: torch.randn(*s, device="cuda", generator=g)
    h = (rn(M, Cc) * 1.5 + 0.3).to(td)
    gamma, beta = 1 + 0.1 * rn(Cc), 0.1 * rn(Cc)
    Wq = (rn(Cc, Cc) / math.sqrt(Cc)).to(td); Wo = (rn(Cc, Cc) / math.sqrt(Cc)).to(td); bo = 0.1 * rn(Cc)
    kv = rn(B * Mk, 2 * Cc).to(td)                       # [K | V] of the projected context, as the engine lays it out
    scale = 1.0 / math.sqrt(D)

    # (a) fp32 torch
    x = h.float()
    q = F.layer_norm(x, (Cc,), gamma, beta, 1e-5) @ Wq.float().t()
    k = kv[:, :Cc].float().view(B, Mk, Hh, D).transpose(1, 2); v = kv[:, Cc:].float().view(B, Mk, Hh, D).transpose(1, 2)
    a = torch.softmax(q.view(B, N, Hh, D).transpose(1, 2) @ k.transpose(-1, -2) * scale, -1) @ v
    ref = x + a.transpose(1, 2).reshape(M, Cc) @ Wo.float().t() + bo

    # (b) the separate ops
    n = torch.empty_like(h); qq = torch.empty_like(h); att = torch.empty_like(h); sep = h.clone()
    ldx.lib.check(L.ldx_op_layernorm(_p(h), Cc, _p(n), Cc, M, Cc, 1e-5, _p(gamma), _p(beta), code, _st()), "ln")
    ldx.lib.check(L.ldx_op_gemm(_p(n), Cc, _p(Wq), M, Cc, Cc, None, None, 0, 1, 0, None, 0, _p(qq), Cc, None, 0, code, _st()), "q")
    ldx.lib.check(L.ldx_op_attention(_p(qq), Cc, _p(kv), 2 * Cc, _p(kv[:, Cc:]), 2 * Cc, _p(att), Cc, B, Hh, N, Mk, D, scale, 0, code, _st()), "attn")
    ldx.lib.check(L.ldx_op_gemm(_p(att), Cc, _p(Wo), M, Cc, Cc, _p(bo), None, 0, 1, 0, _p(sep), Cc, _p(sep), Cc, None, 0, code, _st()), "o")

    fused = h.clone()
    ldx.lib.check(L.ldx_op_xattn_block(_p(fused), Cc, M, N, Cc, Hh, _p(gamma), _p(beta), 1e-5, _p(Wq), _p(Wo), _p(bo), _p(kv), 2 * Cc, _p(kv[:, Cc:]), 2 * Cc,
                                       Mk, scale, code, _st()), "xattn")
    torch.cuda.synchronize()
    # the block's own contribution (what is added to h) is the quantity with signal; h itself dominates the sum
    d_ref, d_sep, d_fused = ref - x, sep.float() - x, fused.float() - x
    r_f, r_s, r_fs = _rel(fused.float(), ref), _rel(sep.float(), ref), _rel(d_fused, d_sep)
    print(f"{dt} B{B} N{N} Mk{Mk}: fused vs torch {r_f:.2e} (separate ops {r_s:.2e}); update: fused vs torch {_rel(d_fused, d_ref):.2e}, fused vs separate {r_fs:.2e}")
    tol = 4e-3 if dt == "bf16" else 6e-4
    assert torch.isfinite(fused).all() and r_f <= tol and r_f <= 1.5 * r_s + 1e-4
    assert _rel(d_fused, d_ref) <= (3e-2 if dt == "bf16" else 4e-3)


def test_xattn_block_refuses_other_shapes(ldx, ldx_lib):
    t = torch.zeros(128, 640, device="cuda", dtype=torch.bfloat16)
    f = torch.zeros(640, device="cuda")
    rc = ldx_lib.ldx_op_xattn_block(_p(t), 640, 128, 128, 640, 8, _p(f), _p(f), 1e-5, _p(t), _p(t), _p(f), _p(t), 640, _p(t), 640, 77, 0.1, 0, _st())
    assert rc != 0
