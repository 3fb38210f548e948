"""Row-block GEMM with a normalisation prologue (csrc/rowgemm.hip, ldx_op_rowgemm) — the C = 320 projections of a transformer block with the
LayerNorm / GroupNorm in front of them computed by the same launch (reference transformer.py:199-209 norm1 + to_q/k/v, to_out + residual;
:361-367 norm + proj_in) — against fp32 torch and against the separate ops it replaces."""
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
@pytest.mark.parametrize("pro,N,res", [(0, 320, True), (0, 320, False), (1, 960, False), (1, 320, True), (2, 320, False)])
@pytest.mark.parametrize("B,HW", [(2, 1024), (1, 384)])
@pytest.mark.parametrize("K", [320, 640])
def test_rowgemm_vs_torch_and_separate_ops(ldx, ldx_lib, dt, pro, N, res, B, HW, K):
    L = ldx_lib
    td, code = DT[dt]
    M = B * HW
    N = N * K // 320                                     # 320 / 960 at K = 320, 640 / 1920 at K = 640
    g = torch.Generator(device="cuda").manual_seed(pro * 100 + N + B + HW + K)
    rn = lambda *s: torch.randn(*s, device="cuda", generator=g)
    X = (rn(M, K) * 1.5 + 0.4).to(td)
    W = (rn(N, K) / math.sqrt(K)).to(td); bias = 0.1 * rn(N)
    gamma, beta = 1 + 0.1 * rn(K), 0.1 * rn(K)
    R = rn(M, N).to(td) if res else None
    x = X.float()
    partial, nchunk, eps = None, 0, 1e-5
    if pro == 0:
        a = x
    elif pro == 1:
        a = F.layer_norm(x, (K,), gamma, beta, eps)
    else:
        eps = 1e-6
        a = F.group_norm(x.view(B, HW, K).transpose(1, 2), 32, gamma, beta, eps).transpose(1, 2).reshape(M, K)
        nchunk = 3                                       # any split of the pixels: ragged chunks
        bounds = [0, HW // 5, HW // 2, HW]
        xs = x.view(B, HW, 32, K // 32)
        partial = torch.stack([torch.stack([xs[:, bounds[c]:bounds[c + 1]].sum((1, 3)), (xs[:, bounds[c]:bounds[c + 1]] ** 2).sum((1, 3))], -1) for c in range(nchunk)], 1).contiguous()
        assert partial.shape == (B, nchunk, 32, 2)
    ref = a @ W.float().t() + bias + (R.float() if res else 0)

    # separate ops: normalisation launch (16-bit output), then the GEMM
    if pro == 0:
        n16 = X
    elif pro == 1:
        n16 = torch.empty_like(X)
        ldx.lib.check(L.ldx_op_layernorm(_p(X), K, _p(n16), K, M, K, eps, _p(gamma), _p(beta), code, _st()), "ln")
    else:
        n16 = torch.empty_like(X)
        ws = torch.zeros(int(L.ldx_op_groupnorm_workspace_floats(B, 32)), device="cuda")
        ldx.lib.check(L.ldx_op_groupnorm(_p(X), K, _p(n16), K, B, HW, K, 32, eps, 0, _p(gamma), _p(beta), _p(ws), code, _st()), "gn")
    sep = torch.empty(M, N, device="cuda", dtype=td)
    ldx.lib.check(L.ldx_op_gemm(_p(n16), K, _p(W), M, N, K, _p(bias), None, 0, 1, 0, _p(R), N, _p(sep), N, None, 0, code, _st()), "gemm")

    Y = torch.zeros(M, N, device="cuda", dtype=td)
    ldx.lib.check(L.ldx_op_rowgemm(_p(X), K, _p(Y), N, M, N, K, _p(W), _p(bias), _p(R), N, pro, _p(gamma), _p(beta), eps, _p(partial), nchunk, HW, code, _st()), "rowgemm")
    torch.cuda.synchronize()
    r_f, r_s = _rel(Y.float(), ref), _rel(sep.float(), ref)
    print(f"{dt} K{K} pro{pro} N{N} res{int(res)} B{B} HW{HW}: fused vs torch {r_f:.2e} (separate ops {r_s:.2e}), fused vs separate {_rel(Y.float(), sep.float()):.2e}")
    tol = 5e-3 if dt == "bf16" else 7e-4
    assert torch.isfinite(Y).all() and r_f <= tol and r_f <= 1.5 * r_s + 1e-4


def test_rowgemm_in_place_residual(ldx, ldx_lib):
    """attn1.to_out: Y = R = the residual stream, X = another buffer."""
    L = ldx_lib
    M, K = 512, 320
    g = torch.Generator(device="cuda").manual_seed(5)
    X = torch.randn(M, K, device="cuda", generator=g).bfloat16(); h = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    W = (torch.randn(K, K, device="cuda", generator=g) / math.sqrt(K)).bfloat16(); bias = torch.randn(K, device="cuda", generator=g)
    ref = h.float() + X.float() @ W.float().t() + bias
    ldx.lib.check(L.ldx_op_rowgemm(_p(X), K, _p(h), K, M, K, K, _p(W), _p(bias), _p(h), K, 0, None, None, 0.0, None, 0, 0, 0, _st()), "rowgemm")
    torch.cuda.synchronize()
    assert _rel(h.float(), ref) <= 5e-3


def test_rowgemm_refuses_other_shapes(ldx, ldx_lib):
    t = torch.zeros(1280, 1280, device="cuda", dtype=torch.bfloat16)
    assert ldx_lib.ldx_op_rowgemm(_p(t), 1280, _p(t), 1280, 128, 1280, 1280, _p(t), None, None, 0, 0, None, None, 0.0, None, 0, 0, 0, _st()) != 0      # K = 1280
    assert ldx_lib.ldx_op_rowgemm(_p(t), 640, _p(t), 1280, 128, 960, 640, _p(t), None, None, 0, 0, None, None, 0.0, None, 0, 0, 0, _st()) != 0        # N % K
