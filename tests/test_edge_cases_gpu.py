"""Degenerate and ragged shapes through the C ABI on a real MI355X: single rows / pixels / keys, tails that are not a
multiple of any tile, zero-length work, and argument validation (every bad call must return an error code with a message,
never crash).  Each kernel is compared with a torch fp32 reference of the same op."""
import ctypes as C
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TD = {"bf16": (torch.bfloat16, 0, 2.5e-2), "f16": (torch.float16, 1, 4e-3)}


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _rel(a, b):
    a, b = a.float(), b.float()
    return float((a - b).norm() / (b.norm() + 1e-20))


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("M,N,K", [(1, 4, 8), (3, 1, 8), (1, 1, 64), (127, 129, 72), (129, 5, 200), (65, 161, 8)])
def test_gemm_tiny_and_ragged(ldx, ldx_lib, dt, M, N, K):
    td, code, tol = TD[dt]
    g = torch.Generator(device="cuda").manual_seed(M * 7 + N * 3 + K)
    A = torch.randn(M, K, device="cuda", generator=g).to(td)
    W = (torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K)).to(td)
    bias = torch.randn(N, device="cuda", generator=g)
    Cf = torch.zeros(M, N, device="cuda")
    Cc = torch.zeros(M, ((N + 3) // 4) * 4 + 4, device="cuda", dtype=td)
    ldx.lib.check(ldx_lib.ldx_op_gemm(_p(A), K, _p(W), M, N, K, _p(bias), None, 0, 1, 0, None, 0, _p(Cc), Cc.shape[1], _p(Cf), N, code, _st()), "gemm")
    ref = A.float() @ W.float().T + bias
    assert _rel(Cf, ref) <= 1e-5 * 50 and _rel(Cc[:, :N], ref) <= tol
    assert torch.all(Cc[:, N:] == 0)


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("B,H,W,Cin,Cout,stride", [(1, 1, 1, 64, 64, 1), (1, 2, 3, 64, 8, 1), (2, 1, 7, 128, 64, 2), (1, 5, 1, 64, 4, 2), (3, 3, 3, 64, 160, 1)])
def test_conv_tiny_images(ldx, ldx_lib, dt, B, H, W, Cin, Cout, stride):
    td, code, tol = TD[dt]
    g = torch.Generator(device="cuda").manual_seed(B + H * 5 + W * 11 + Cin + Cout)
    Ho, Wo = (H + stride - 1) // stride if stride == 2 else H, (W + stride - 1) // stride if stride == 2 else W
    X = torch.randn(B, H, W, Cin, device="cuda", generator=g).to(td)
    Wt = (torch.randn(Cout, Cin, 3, 3, device="cuda", generator=g) / math.sqrt(9 * Cin)).to(td)
    Wp = Wt.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous()
    bias = torch.randn(Cout, device="cuda", generator=g)
    Y = torch.zeros(B * Ho * Wo, Cout, device="cuda", dtype=td)
    ldx.lib.check(ldx_lib.ldx_op_conv3x3(_p(X), Cin, _p(Wp), B, H, W, Cin, Cout, stride, Ho, Wo, 0, _p(bias), None, 0, None, 0, _p(Y), Cout, code, _st()), "conv")
    ref = F.conv2d(X.float().permute(0, 3, 1, 2), Wt.float(), bias, stride=stride, padding=1)
    assert ref.shape[-2:] == (Ho, Wo)
    assert _rel(Y, ref.permute(0, 2, 3, 1).reshape(-1, Cout)) <= tol


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("B,H,Nq,Mk,D,causal", [(1, 1, 1, 1, 8, 0), (1, 2, 1, 77, 40, 0), (2, 1, 5, 3, 64, 0), (1, 3, 65, 65, 16, 1), (1, 1, 130, 1, 160, 0),
                                                (1, 2, 63, 129, 80, 0), (1, 1, 7, 7, 8, 1)])
def test_attention_tiny_and_ragged(ldx, ldx_lib, dt, B, H, Nq, Mk, D, causal):
    td, code, tol = TD[dt]
    g = torch.Generator(device="cuda").manual_seed(Nq * 13 + Mk * 5 + D)
    Cc = H * D
    q = torch.randn(B, Nq, Cc, device="cuda", generator=g).to(td)
    k = torch.randn(B, Mk, Cc, device="cuda", generator=g).to(td)
    v = torch.randn(B, Mk, Cc, device="cuda", generator=g).to(td)
    o = torch.zeros(B, Nq, Cc, device="cuda", dtype=td)
    sc = 1.0 / math.sqrt(D)
    ldx.lib.check(ldx_lib.ldx_op_attention(_p(q), Cc, _p(k), Cc, _p(v), Cc, _p(o), Cc, B, H, Nq, Mk, D, sc, causal, code, _st()), "attn")
    qf, kf, vf = (t.float().reshape(B, -1, H, D).transpose(1, 2) for t in (q, k, v))
    s = qf @ kf.transpose(-1, -2) * sc
    if causal:
        s = s.masked_fill(torch.triu(torch.ones(Nq, Mk, device="cuda", dtype=torch.bool), 1), float("-inf"))
    ref = (torch.softmax(s, -1) @ vf).transpose(1, 2).reshape(B, Nq, Cc)
    assert _rel(o, ref) <= 2 * tol


@pytest.mark.parametrize("dt", ["bf16", "f16"])
def test_norms_single_row_and_pixel(ldx, ldx_lib, dt):
    td, code, tol = TD[dt]
    g = torch.Generator(device="cuda").manual_seed(2)
    for rows, Cn in ((1, 64), (3, 320), (5, 3072), (17, 4096)):
        X = torch.randn(rows, Cn, device="cuda", generator=g).to(td)
        gm, bt = torch.randn(Cn, device="cuda", generator=g), torch.randn(Cn, device="cuda", generator=g)
        Y = torch.zeros_like(X)
        if Cn <= 4096:
            ldx.lib.check(ldx_lib.ldx_op_layernorm(_p(X), Cn, _p(Y), Cn, rows, Cn, 1e-5, _p(gm), _p(bt), code, _st()), "ln")
            assert _rel(Y, F.layer_norm(X.float(), (Cn,), gm, bt, 1e-5)) <= tol
    for B, HW, Cn in ((1, 1, 64), (2, 3, 320), (1, 7, 2560)):
        X = torch.randn(B, HW, Cn, device="cuda", generator=g).to(td)
        gm, bt = torch.randn(Cn, device="cuda", generator=g), torch.randn(Cn, device="cuda", generator=g)
        Y = torch.zeros_like(X)
        ws = torch.zeros(ldx_lib.ldx_op_groupnorm_workspace_floats(B, 32), device="cuda")
        ldx.lib.check(ldx_lib.ldx_op_groupnorm(_p(X), Cn, _p(Y), Cn, B, HW, Cn, 32, 1e-5, 0, _p(gm), _p(bt), _p(ws), code, _st()), "gn")
        ref = F.group_norm(X.float().permute(0, 2, 1), 32, gm, bt, 1e-5).permute(0, 2, 1)
        assert _rel(Y, ref) <= tol


def test_zero_length_and_bad_arguments(ldx, ldx_lib):
    L = ldx_lib
    x = torch.randn(16, device="cuda")
    # n = 0 is legal and a no-op
    y = x.clone()
    assert L.ldx_sampler_step(0, _p(y), _p(x), _p(x), None, 0, 7.0, 1.0, -0.1, _st()) == 0 and torch.equal(x, y)
    # every malformed call reports an error code and leaves a message
    bad = [
        lambda: L.ldx_sampler_step(9, _p(y), _p(x), _p(x), None, 16, 7.0, 1.0, -0.1, _st()),
        lambda: L.ldx_sampler_step(0, None, _p(x), _p(x), None, 16, 7.0, 1.0, -0.1, _st()),
        lambda: L.ldx_op_gemm(_p(x), 8, _p(x), 2, 2, 7, None, None, 0, 1, 0, None, 0, _p(y), 4, None, 0, 0, _st()),          # K % 8
        lambda: L.ldx_op_attention(_p(x), 8, _p(x), 8, _p(x), 8, _p(y), 8, 1, 1, 1, 1, 168, 1.0, 0, 0, _st()),              # D > 160
        lambda: L.ldx_op_attention(_p(x), 8, _p(x), 8, _p(x), 8, _p(y), 8, 1, 1, 1, 0, 8, 1.0, 0, 0, _st()),                # no keys
        lambda: L.ldx_bislerp_pass(_p(x), _p(y), 1, 17, 1, 1, 1, 1, _p(x), _p(x), _p(x), _st()),                            # C > 16
        lambda: L.ldx_tile_blend(_p(x), 4, 4, _p(y), _p(y), 2, 2, 1, 0, 0, 1, _st()),                                       # tile outside output
        lambda: L.ldx_unet_denoise(None, _p(x), _p(x), _p(x), 1, 8, 8, 77, _p(y), _st()),                                   # null engine
    ]
    for call in bad:
        assert call() != 0
        assert len(L.ldx_last_error()) > 0
    # an engine of the wrong kind is rejected, not dereferenced
    cfg = ldx.CLIPConfig.tiny()
    sd = ldx.weights.synth_state_dict(ldx.weights.clip_state_dict_spec(cfg), seed=1)
    eng = ldx.CLIPTextEngine(cfg, sd, device=0, dtype="f16")
    assert L.ldx_vae_decode(eng._h, _p(x), 1, 4, 4, _p(y), _st()) != 0
    assert L.ldx_t5_encode(eng._h, _p(x), 1, 4, _p(x), _p(y), _st()) != 0
    assert L.ldx_flux_fbcache(eng._h, C.c_float(0.1)) != 0
