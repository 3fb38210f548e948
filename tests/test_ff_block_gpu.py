"""The fused feed-forward sub-block (csrc/ff_block.hip, ldx_op_ff_block; reference transformer.py:19-70, 240-244 + Activation.py:6-31):
LayerNorm + GEGLU projection + down projection + bias + residual in one launch, against (a) fp32 torch on the same 16-bit inputs and
(b) the three separate ops it replaces (ldx_op_layernorm, ldx_op_gemm with the GEGLU epilogue, ldx_op_gemm with residual)."""
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


def geglu_rows(inner):
    """engine row order of the GEGLU projection (engine.cpp mk_xf): slab s = value rows 32 s .. 32 s + 31, then gate rows inner + 32 s .."""
    r = torch.arange(2 * inner)
    slab, within = r // 64, r % 64
    return torch.where(within < 32, slab * 32 + within, inner + slab * 32 + (within - 32))


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("M", [128, 1000, 32768])
def test_ff_block_vs_torch_and_separate_ops(ldx, ldx_lib, dt, M):
    L = ldx_lib
    td, code = DT[dt]
    Cc, inner = 320, 1280
    g = torch.Generator(device="cuda").manual_seed(M)
    rn = lambda *s: torch.randn(*s, device="cuda", generator=g)
    h = (rn(M, Cc) * 1.5 + 0.3).to(td)
    gamma, beta = 1 + 0.1 * rn(Cc), 0.1 * rn(Cc)
    W1 = (rn(2 * inner, Cc) / math.sqrt(Cc)).to(td); b1 = 0.1 * rn(2 * inner)           # reference order: [value rows | gate rows]
    W2 = (rn(Cc, inner) / math.sqrt(inner)).to(td); b2 = 0.1 * rn(Cc)
    perm = geglu_rows(inner).cuda()
    W1e, b1e = W1[perm].contiguous(), b1[perm].contiguous()

    x = h.float()
    t = F.layer_norm(x, (Cc,), gamma, beta, 1e-5) @ W1.float().t() + b1
    ref = x + (t[:, :inner] * F.gelu(t[:, inner:])) @ W2.float().t() + b2

    n = torch.empty_like(h); f = torch.empty(M, inner, device="cuda", dtype=td); sep = h.clone()
    ldx.lib.check(L.ldx_op_layernorm(_p(h), Cc, _p(n), Cc, M, Cc, 1e-5, _p(gamma), _p(beta), code, _st()), "ln")
    ldx.lib.check(L.ldx_op_gemm(_p(n), Cc, _p(W1e), M, 2 * inner, Cc, _p(b1e), None, 0, 1, 1, None, 0, _p(f), inner, None, 0, code, _st()), "ff1")
    ldx.lib.check(L.ldx_op_gemm(_p(f), inner, _p(W2), M, Cc, inner, _p(b2), None, 0, 1, 0, _p(sep), Cc, _p(sep), Cc, None, 0, code, _st()), "ff2")

    fused = h.clone()
    ldx.lib.check(L.ldx_op_ff_block(_p(fused), Cc, M, Cc, inner, _p(gamma), _p(beta), 1e-5, _p(W1e), _p(b1e), _p(W2), _p(b2), code, _st()), "ffblock")
    torch.cuda.synchronize()
    d_ref, d_sep, d_fused = ref - x, sep.float() - x, fused.float() - x
    r_f, r_s = _rel(fused.float(), ref), _rel(sep.float(), ref)
    print(f"{dt} M{M}: fused vs torch {r_f:.2e} (separate ops {r_s:.2e}); update: fused vs torch {_rel(d_fused, d_ref):.2e}, fused vs separate {_rel(d_fused, d_sep):.2e}")
    tol = 4e-3 if dt == "bf16" else 6e-4
    assert torch.isfinite(fused).all() and r_f <= tol and r_f <= 1.5 * r_s + 1e-4
    assert _rel(d_fused, d_ref) <= (2e-2 if dt == "bf16" else 3e-3)


def test_ff_block_refuses_other_shapes(ldx, ldx_lib):
    t = torch.zeros(128, 640, device="cuda", dtype=torch.bfloat16)
    f = torch.zeros(5120, device="cuda")
    assert ldx_lib.ldx_op_ff_block(_p(t), 640, 128, 640, 2560, _p(f), _p(f), 1e-5, _p(t), _p(f), _p(t), _p(f), 0, _st()) != 0
