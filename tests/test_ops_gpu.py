"""Per-kernel parity on a real MI355X: every HIP kernel is called through the C ABI (libldx.so) and compared
with a plain PyTorch fp32 reference of the same op on the same 16-bit inputs.

Tolerances: outputs are stored in 16-bit, accumulation is fp32, so the bound is one output rounding
(bf16: 2^-8 relative, fp16: 2^-11) plus reduction-order noise.  We assert on rel-L2 (<= 4e-3 bf16 / 6e-4 fp16)
and on max-abs relative to the output scale (<= 2e-2 bf16 / 4e-3 fp16); attention adds the bf16 rounding of P.
"""
import ctypes as C
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DT = {"bf16": (torch.bfloat16, 0), "f16": (torch.float16, 1)}
TOL = {"bf16": (4e-3, 2e-2), "f16": (6e-4, 4e-3)}


def _check(got, ref, dt, scale=1.0, what=""):
    got, ref = got.float(), ref.float()
    rel = float((got - ref).norm() / (ref.norm() + 1e-20))
    mx = float((got - ref).abs().max() / (ref.abs().max() + 1e-20))
    r, m = TOL[dt]
    assert math.isfinite(rel) and rel <= r * scale and mx <= m * scale, f"{what}: rel-L2 {rel:.3e} max {mx:.3e}"


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


@pytest.fixture(scope="module")
def L(ldx_lib):
    assert torch.cuda.is_available()
    return ldx_lib


def _st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("M,N,K,lda_extra", [(300, 200, 128, 0), (1024, 320, 320, 64), (154, 640, 768, 0), (32768, 320, 320, 0), (513, 4, 2880 // 64 * 64, 0),
                                              (2048, 1280, 1280, 0), (512, 1280, 2560, 0), (200, 96, 96, 0), (130, 64, 200, 8)])   # last three take the split-K path
def test_gemm(L, ldx, dt, M, N, K, lda_extra):
    td, code = DT[dt]
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    lda = K + lda_extra
    A = torch.randn(M, lda, device="cuda", generator=g).to(td)
    W = (torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K)).to(td)
    bias = torch.randn(N, device="cuda", generator=g)
    nb = 4
    rpb = (M + nb - 1) // nb
    rowvec = torch.randn(nb, N + 8, device="cuda", generator=g)
    R = torch.randn(M, N + 16, device="cuda", generator=g).to(td)
    Cc = torch.zeros(M, N + 8, device="cuda", dtype=td)
    Cf = torch.zeros(M, N, device="cuda", dtype=torch.float32)
    ldx.lib.check(L.ldx_op_gemm(_p(A), lda, _p(W), M, N, K, _p(bias), _p(rowvec), N + 8, rpb, 0, _p(R), N + 16,
                                _p(Cc), N + 8, _p(Cf), N, code, _st()), "gemm")
    torch.cuda.synchronize()
    ref = A[:, :K].float() @ W.float().T + bias + rowvec[torch.arange(M, device="cuda") // rpb, :N] + R[:, :N].float()
    _check(Cc[:, :N], ref, dt, what="gemm C16")
    _check(Cf, ref, "f16", what="gemm Cf32")            # fp32 output: no output rounding at all
    assert torch.all(Cc[:, N:] == 0)                    # ldc padding untouched


@pytest.mark.parametrize("dt", ["bf16", "f16"])
def test_gemm_geglu(L, ldx, dt):
    td, code = DT[dt]
    g = torch.Generator(device="cuda").manual_seed(3)
    M, C = 700, 128
    inner = 4 * C
    A = torch.randn(M, C, device="cuda", generator=g).to(td)
    Wfull = (torch.randn(2 * inner, C, device="cuda", generator=g) / math.sqrt(C)).to(td)
    bfull = torch.randn(2 * inner, device="cuda", generator=g)
    # slab interleave as engine.cpp::mk_xf does: 64-row slabs = 32 value rows then the 32 matching gate rows
    idx = torch.arange(2 * inner, device="cuda")
    slab, within = idx // 64, idx % 64
    src = torch.where(within < 32, slab * 32 + within, inner + slab * 32 + (within - 32))
    Wp, bp = Wfull[src].contiguous(), bfull[src].contiguous()
    out = torch.zeros(M, inner, device="cuda", dtype=td)
    ldx.lib.check(L.ldx_op_gemm(_p(A), C, _p(Wp), M, 2 * inner, C, _p(bp), None, 0, 1, 1, None, 0, _p(out), inner, None, 0, code, _st()), "geglu")
    torch.cuda.synchronize()
    y = A.float() @ Wfull.float().T + bfull
    a, gate = y.chunk(2, dim=-1)
    _check(out, a * F.gelu(gate), dt, what="geglu")


@pytest.mark.parametrize("M,C", [(8192, 640), (2048, 1280), (2048 + 96, 1280)])
def test_gemm_geglu_unet_ff_shapes(L, ldx, M, C):
    """The FF up-projections of the 64^2 / 32^2 levels (transformer.py:19-70, GEGLU cond/Activation.py:6-31) at their production shapes, where the
    planner takes 256-wide ping-pong tiles (two 64-column GEGLU slabs per wave, round 6) — bias, ragged last row tile, value / gate pairing."""
    td, code = DT["bf16"]
    g = torch.Generator(device="cuda").manual_seed(5)
    inner = 4 * C
    A = torch.randn(M, C, device="cuda", generator=g).to(td)
    Wfull = (torch.randn(2 * inner, C, device="cuda", generator=g) / math.sqrt(C)).to(td)
    bfull = torch.randn(2 * inner, device="cuda", generator=g)
    idx = torch.arange(2 * inner, device="cuda")
    slab, within = idx // 64, idx % 64
    src = torch.where(within < 32, slab * 32 + within, inner + slab * 32 + (within - 32))
    Wp, bp = Wfull[src].contiguous(), bfull[src].contiguous()
    out = torch.zeros(M, inner, device="cuda", dtype=td)
    ldx.lib.check(L.ldx_op_gemm(_p(A), C, _p(Wp), M, 2 * inner, C, _p(bp), None, 0, 1, 1, None, 0, _p(out), inner, None, 0, code, _st()), "geglu")
    torch.cuda.synchronize()
    y = A.float() @ Wfull.float().T + bfull
    a, gate = y.chunk(2, dim=-1)
    _check(out, a * F.gelu(gate), "bf16", what=f"geglu M{M} C{C}")


CONV_CASES = [
    # B, Hin, Win, Cin, Cout, stride, Hout, Wout, resize, ldx_extra
    (2, 16, 16, 64, 128, 1, 16, 16, 0, 0),
    (2, 17, 13, 128, 64, 2, 9, 7, 0, 64),       # odd sizes, stride 2 -> ceil
    (1, 8, 8, 64, 64, 1, 16, 16, 1, 0),         # nearest 2x + conv (Upsample1)
    (1, 5, 7, 64, 64, 1, 9, 13, 1, 0),          # resize to an odd skip shape
    (2, 64, 64, 320, 320, 1, 64, 64, 0, 0),
    (1, 16, 16, 64, 4, 1, 16, 16, 0, 0),        # Cout = 4 (UNet out conv)
    (2, 16, 16, 1280, 1280, 1, 16, 16, 0, 0),   # split-K path (40 tiles, 180 K-tiles)
    (2, 33, 31, 640, 640, 2, 17, 16, 0, 0),     # split-K + stride 2
]


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv3x3(L, ldx, dt, case):
    td, code = DT[dt]
    B, Hin, Win, Cin, Cout, stride, Hout, Wout, resize, extra = case
    g = torch.Generator(device="cuda").manual_seed(sum(case))
    ld = Cin + extra
    X = torch.randn(B, Hin, Win, ld, device="cuda", generator=g).to(td)               # NHWC, strided channels
    Wt = (torch.randn(Cout, Cin, 3, 3, device="cuda", generator=g) / math.sqrt(9 * Cin)).to(td)
    Wp = Wt.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous()                  # [Cout][ky][kx][Cin]
    bias = torch.randn(Cout, device="cuda", generator=g)
    rowvec = torch.randn(B, Cout, device="cuda", generator=g)
    R = torch.randn(B * Hout * Wout, Cout, device="cuda", generator=g).to(td)
    Y = torch.zeros(B * Hout * Wout, Cout, device="cuda", dtype=td)
    ldx.lib.check(L.ldx_op_conv3x3(_p(X), ld, _p(Wp), B, Hin, Win, Cin, Cout, stride, Hout, Wout, resize, _p(bias),
                                   _p(rowvec), Cout, _p(R), Cout, _p(Y), Cout, code, _st()), "conv")
    torch.cuda.synchronize()
    xin = X[..., :Cin].float().permute(0, 3, 1, 2)
    if resize:
        xin = F.interpolate(xin, size=(Hout, Wout), mode="nearest")
    ref = F.conv2d(xin, Wt.float(), bias, stride=stride, padding=1) + rowvec[:, :, None, None]
    assert ref.shape[-2:] == (Hout, Wout)
    ref = ref.permute(0, 2, 3, 1).reshape(B * Hout * Wout, Cout) + R.float()
    _check(Y, ref, dt, what=f"conv {case}")


PATCH_CASES = [
    # B, Hin, Win, Cin, Cout, Hout, Wout, resize, ldx_extra, residual   (conv_patch.hip: Cout 32 / 64 / 128, 32 x 16-pixel tiles, >= 256 of them)
    (1, 256, 512, 64, 32, 256, 512, 0, 128, 0),      # ESRGAN RDB conv1: reads 64 of 192 columns; one tile per workgroup
    (1, 256, 512, 192, 64, 256, 512, 0, 0, 1),       # RDB conv5: 6 channel chunks, residual
    (2, 128, 512, 128, 32, 128, 512, 0, 32, 0),      # 4 chunks, two images
    (1, 128, 256, 64, 64, 256, 512, 1, 0, 0),        # nearest x2 + conv (upconv_block)
    (1, 150, 111, 64, 64, 400, 352, 1, 0, 1),        # non-integer resize ratios, 275 tiles (uneven tiles per workgroup)
    (1, 256, 512, 128, 128, 256, 512, 0, 0, 1),      # VAE last level 128 -> 128
    (1, 272, 544, 64, 128, 272, 544, 0, 64, 0),      # 289 tiles: some workgroups walk two
    (1, 512, 1024, 64, 32, 512, 1024, 0, 0, 0),      # 1024 tiles: four per workgroup, the load stream crosses three tile boundaries
    (1, 256, 512, 64, 3, 256, 512, 0, 0, 0),         # conv_last / conv_out: 3 channels (weight rows 3 .. 15 read as zeros), element-wise tail
    (1, 256, 512, 128, 4, 256, 512, 0, 64, 0),       # UNet out conv shape: 4 channels
]


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("case", PATCH_CASES)
def test_conv3x3_patch_resident(L, ldx, dt, case):
    """The narrow-output 3x3 convs that launch_gemm routes to conv_patch.hip (USDU_util.py:36-98 conv_block / :101-128 upconv_block shapes):
    borders (zero padding from the buffer bounds check), halo sharing between tiles, the resize gather, strided input views."""
    td, code = DT[dt]
    B, Hin, Win, Cin, Cout, Hout, Wout, resize, extra, res = case
    g = torch.Generator(device="cuda").manual_seed(sum(case))
    ld = Cin + extra
    X = torch.randn(B, Hin, Win, ld, device="cuda", generator=g).to(td)
    Wt = (torch.randn(Cout, Cin, 3, 3, device="cuda", generator=g) / math.sqrt(9 * Cin)).to(td)
    Wp = Wt.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous()
    bias = torch.randn(Cout, device="cuda", generator=g)
    R = torch.randn(B * Hout * Wout, Cout + 8, device="cuda", generator=g).to(td) if res else None
    Y = torch.zeros(B * Hout * Wout, Cout + 16, device="cuda", dtype=td)
    ldx.lib.check(L.ldx_op_conv3x3(_p(X), ld, _p(Wp), B, Hin, Win, Cin, Cout, 1, Hout, Wout, resize, _p(bias),
                                   None, 0, _p(R), Cout + 8, _p(Y), Cout + 16, code, _st()), "conv")
    torch.cuda.synchronize()
    xin = X[..., :Cin].float().permute(0, 3, 1, 2)
    if resize:
        xin = F.interpolate(xin, size=(Hout, Wout), mode="nearest")
    ref = F.conv2d(xin, Wt.float(), bias, padding=1).permute(0, 2, 3, 1).reshape(B * Hout * Wout, Cout)
    if res:
        ref = ref + R[:, :Cout].float()
    _check(Y[:, :Cout], ref, dt, what=f"patch conv {case}")
    assert float(Y[:, Cout:].abs().max()) == 0.0          # nothing written beyond the Cout columns of the view


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("B,H,W,Cin,Cin2,Cout", [(2, 16, 16, 1280, 2560, 1280),     # split-K 11: several splits start inside the second segment
                                                 (2, 32, 32, 640, 1280, 640), (1, 24, 20, 128, 64, 192), (2, 64, 64, 320, 640, 320)])
def test_conv3x3_with_fused_skip(L, ldx, dt, B, H, W, Cin, Cin2, Cout):
    """conv2(h) + skip_connection(x) of ResBlock1 as one implicit GEMM over K = 9*Cin + Cin2 (GemmArgs::A2)."""
    td, code = DT[dt]
    g = torch.Generator(device="cuda").manual_seed(B + H + Cin + Cin2)
    X = torch.randn(B, H, W, Cin + 64, device="cuda", generator=g).to(td)
    X2 = torch.randn(B, H, W, Cin2 + 8, device="cuda", generator=g).to(td)
    W3 = (torch.randn(Cout, Cin, 3, 3, device="cuda", generator=g) / math.sqrt(9 * Cin)).to(td)
    W1 = (torch.randn(Cout, Cin2, device="cuda", generator=g) / math.sqrt(Cin2)).to(td)
    Wp = torch.cat([W3.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin), W1], dim=1).contiguous()
    bias = torch.randn(Cout, device="cuda", generator=g)
    Y = torch.zeros(B * H * W, Cout, device="cuda", dtype=td)
    ldx.lib.check(L.ldx_op_conv3x3_skip(_p(X), Cin + 64, _p(X2), Cin2 + 8, Cin2, _p(Wp), B, H, W, Cin, Cout, _p(bias), _p(Y), Cout, code, _st()),
                  "conv+skip")
    torch.cuda.synchronize()
    ref = F.conv2d(X[..., :Cin].float().permute(0, 3, 1, 2), W3.float(), bias, padding=1)
    ref = ref + F.conv2d(X2[..., :Cin2].float().permute(0, 3, 1, 2), W1.float()[:, :, None, None])
    _check(Y, ref.permute(0, 2, 3, 1).reshape(B * H * W, Cout), dt, what="conv+skip")


def test_conv3x3_image_over_2gib(L, ldx):
    """VAE decode at 2048^2 (config 5): one NHWC input image of 2048*2048*256 bf16 = 2 GiB.  The conv loader's buffer
    window is per tile, so rows whose byte offset exceeds 2^31 must still be right: check bands at the top, around the
    2 GiB mark and at the bottom against torch's conv on the same rows."""
    td, code = DT["bf16"]
    B, H, W, Cin, Cout = 1, 2048, 2048, 256, 64
    g = torch.Generator(device="cuda").manual_seed(5)
    X = torch.randn(B, H, W, Cin, device="cuda", generator=g, dtype=torch.float32).to(td)
    Wt = (torch.randn(Cout, Cin, 3, 3, device="cuda", generator=g) / math.sqrt(9 * Cin)).to(td)
    Wp = Wt.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous()
    bias = torch.randn(Cout, device="cuda", generator=g)
    Y = torch.zeros(B * H * W, Cout, device="cuda", dtype=td)
    ldx.lib.check(L.ldx_op_conv3x3(_p(X), Cin, _p(Wp), B, H, W, Cin, Cout, 1, H, W, 0, _p(bias), None, 0, None, 0,
                                   _p(Y), Cout, code, _st()), "conv")
    torch.cuda.synchronize()
    Y = Y.view(H, W, Cout)
    for r0, r1 in ((0, 4), (1022, 1027), (1530, 1534), (2044, 2048)):
        lo, hi = max(r0 - 1, 0), min(r1 + 1, H)
        band = X[0, lo:hi].float().permute(2, 0, 1)[None]                        # [1, Cin, rows, W]
        band = F.pad(band, (0, 0, 1 if r0 == 0 else 0, 1 if r1 == H else 0))
        ref = F.conv2d(F.pad(band, (1, 1, 0, 0)), Wt.float(), bias)[0].permute(1, 2, 0)
        assert ref.shape[0] == r1 - r0
        _check(Y[r0:r1].reshape(-1, Cout), ref.reshape(-1, Cout), "bf16", what=f"conv rows {r0}:{r1}")


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("B,HW,C,extra,silu,eps", [(2, 256, 64, 0, 1, 1e-5), (2, 1024, 320, 64, 1, 1e-5), (1, 300, 2560, 0, 0, 1e-6),
                                                   (2, 4096, 192, 0, 1, 1e-5), (1, 16384, 320, 0, 1, 1e-5), (3, 77, 1920, 0, 1, 1e-5)])
def test_groupnorm(L, ldx, dt, B, HW, C, extra, silu, eps):
    td, code = DT[dt]
    g = torch.Generator(device="cuda").manual_seed(C + HW)
    ld = C + extra
    X = (torch.randn(B, HW, ld, device="cuda", generator=g) * 2.0 + 0.7).to(td)
    gamma = 1 + 0.1 * torch.randn(C, device="cuda", generator=g)
    beta = 0.1 * torch.randn(C, device="cuda", generator=g)
    Y = torch.zeros(B, HW, C, device="cuda", dtype=td)
    ws = torch.zeros(L.ldx_op_groupnorm_workspace_floats(B, 32), device="cuda")
    ldx.lib.check(L.ldx_op_groupnorm(_p(X), ld, _p(Y), C, B, HW, C, 32, eps, silu, _p(gamma), _p(beta), _p(ws), code, _st()), "gn")
    torch.cuda.synchronize()
    ref = F.group_norm(X[..., :C].float().permute(0, 2, 1), 32, gamma, beta, eps)
    if silu:
        ref = F.silu(ref)
    _check(Y, ref.permute(0, 2, 1), dt, what="groupnorm")


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("rows,C", [(1000, 320), (77, 768), (513, 1280), (9, 64)])
def test_layernorm(L, ldx, dt, rows, C):
    td, code = DT[dt]
    g = torch.Generator(device="cuda").manual_seed(rows + C)
    X = (torch.randn(rows, C, device="cuda", generator=g) * 3 - 1).to(td)
    gamma = 1 + 0.1 * torch.randn(C, device="cuda", generator=g)
    beta = 0.1 * torch.randn(C, device="cuda", generator=g)
    Y = torch.zeros(rows, C, device="cuda", dtype=td)
    ldx.lib.check(L.ldx_op_layernorm(_p(X), C, _p(Y), C, rows, C, 1e-5, _p(gamma), _p(beta), code, _st()), "ln")
    torch.cuda.synchronize()
    _check(Y, F.layer_norm(X.float(), (C,), gamma, beta, 1e-5), dt, what="layernorm")


ATTN_CASES = [
    # B, H, Nq, Mk, D, causal, fused_qkv
    (2, 8, 256, 256, 8, 0, 1), (2, 8, 200, 200, 16, 0, 1), (1, 8, 1024, 1024, 32, 0, 1), (2, 8, 1024, 1024, 40, 0, 1),
    (2, 8, 4096, 77, 40, 0, 0), (1, 8, 300, 154, 80, 0, 0), (2, 8, 256, 256, 160, 0, 1), (2, 12, 77, 77, 64, 1, 1),
    (1, 8, 4096, 4096, 40, 0, 1), (1, 2, 130, 231, 64, 0, 0), (1, 4, 200, 200, 64, 1, 1),
    # >= 512 workgroups of 256 queries at D = 40: the 32x32x16 kernel (attn32_kernel), ragged key and query tails
    (4, 16, 2048, 333, 40, 0, 0), (8, 8, 2000, 2048, 40, 0, 0), (2, 32, 2048, 64, 40, 0, 0),
    # >= 64 workgroups of 128 queries at D = 80 / 160: the generic 32x32x16 kernel (attn32g_kernel)
    (2, 8, 1024, 1024, 160, 0, 1), (2, 8, 4096, 77, 80, 0, 0), (2, 16, 1000, 333, 80, 0, 0), (4, 8, 513, 1024, 160, 0, 0), (2, 8, 4096, 4096, 80, 0, 1),
    (1, 24, 1100, 1100, 128, 0, 1), (2, 12, 700, 130, 128, 0, 0),       # D = 128 (Flux): VALU-denominator variant
]


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("case", ATTN_CASES)
def test_attention(L, ldx, dt, case):
    td, code = DT[dt]
    B, H, Nq, Mk, D, causal, fused = case
    g = torch.Generator(device="cuda").manual_seed(sum(case))
    C_ = H * D
    if fused:
        qkv = torch.randn(B, Nq, 3 * C_, device="cuda", generator=g).to(td)
        q, k, v = qkv[..., :C_], qkv[..., C_:2 * C_], qkv[..., 2 * C_:]
        ldq = ldk = ldv = 3 * C_
    else:
        q = torch.randn(B, Nq, C_, device="cuda", generator=g).to(td)
        kv = torch.randn(B, Mk, 2 * C_, device="cuda", generator=g).to(td)
        k, v = kv[..., :C_], kv[..., C_:]
        ldq, ldk, ldv = C_, 2 * C_, 2 * C_
    O_ = torch.zeros(B, Nq, C_, device="cuda", dtype=td)
    scale = 1.0 / math.sqrt(D)
    ldx.lib.check(L.ldx_op_attention(_p(q), ldq, _p(k), ldk, _p(v), ldv, _p(O_), C_, B, H, Nq, Mk, D, scale, causal, code, _st()), "attn")
    torch.cuda.synchronize()
    qf, kf, vf = (t.float().reshape(B, -1, H, D).transpose(1, 2) for t in (q, k, v))
    s = (qf @ kf.transpose(-1, -2)) * scale
    if causal:
        s = s.masked_fill(torch.triu(torch.ones(Nq, Mk, device="cuda", dtype=torch.bool), 1), float("-inf"))
    ref = (torch.softmax(s, dim=-1) @ vf).transpose(1, 2).reshape(B, Nq, C_)
    _check(O_, ref, dt, scale=2.0, what=f"attention {case}")


@pytest.mark.parametrize("shape", [(1, 2, 512, 40), (4, 16, 2048, 40)])      # second shape: 256 workgroups of 512 queries -> attn32ap_kernel (round 3)
def test_attention_online_softmax_rescale(L, ldx, shape):
    """Force the running-max rescale: one key late in the sequence dominates every row (cdna guide rule 26)."""
    td, code = DT["bf16"]
    B, H, N, D = shape
    g = torch.Generator(device="cuda").manual_seed(9)
    q = torch.randn(B, N, H * D, device="cuda", generator=g)
    k = torch.randn(B, N, H * D, device="cuda", generator=g)
    v = torch.randn(B, N, H * D, device="cuda", generator=g)
    k[:, 300] = q[:, 17] * 6.0          # spike in the 5th key block
    k[:, 450] = -q[:, 17] * 6.0
    q, k, v = q.to(td), k.to(td), v.to(td)
    O_ = torch.zeros(B, N, H * D, device="cuda", dtype=td)
    scale = 1.0 / math.sqrt(D)
    ldx.lib.check(L.ldx_op_attention(_p(q), H * D, _p(k), H * D, _p(v), H * D, _p(O_), H * D, B, H, N, N, D, scale, 0, code, _st()), "attn")
    qf, kf, vf = (t.double().reshape(B, N, H, D).transpose(1, 2) for t in (q, k, v))
    ref = (torch.softmax(qf @ kf.transpose(-1, -2) * scale, -1) @ vf).transpose(1, 2).reshape(B, N, H * D)
    _check(O_, ref, "bf16", scale=2.0, what="attention rescale")


@pytest.mark.parametrize("dt", ["bf16", "f16"])
def test_skinny(L, ldx, dt):
    td, code = DT[dt]
    g = torch.Generator(device="cuda").manual_seed(5)
    M, N, K = 6, 1000, 1280
    x = torch.randn(M, K, device="cuda", generator=g)
    W = (torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K)).to(td)
    b = torch.randn(N, device="cuda", generator=g)
    out = torch.zeros(M, N, device="cuda")
    ldx.lib.check(L.ldx_op_skinny(_p(x), K, _p(W), _p(b), _p(out), N, M, N, K, 1, 1, code, _st()), "skinny")
    torch.cuda.synchronize()
    ref = F.silu(F.linear(F.silu(x), W.float(), b))
    assert float((out - ref).norm() / ref.norm()) < 1e-5


def test_sampler_step_and_bilinear(L, ldx):
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(2, 4, 32, 24, device="cuda", generator=g)
    du, dc = torch.randn_like(x), torch.randn_like(x)
    for kind, c0, c1 in ((0, 3.1, -0.7), (1, 0.8, -0.23)):
        xx, d = x.clone(), torch.empty_like(x)
        ldx.lib.check(L.ldx_sampler_step(kind, _p(xx), _p(du), _p(dc), _p(d), x.numel(), 7.0, c0, c1, _st()), "step")
        dref = torch.lerp(du, dc, 7.0)
        xref = x + ((x - dref) / c0) * c1 if kind == 0 else c0 * x - c1 * dref
        assert torch.allclose(d, dref, rtol=1e-6, atol=1e-6) and torch.allclose(xx, xref, rtol=1e-6, atol=1e-6)
    for size in ((16, 8), (64, 48), (8, 8), (33, 25)):
        out = torch.empty(2, 4, *size, device="cuda")
        ldx.lib.check(L.ldx_bilinear(_p(x), _p(out), 8, 32, 24, size[0], size[1], _st()), "bilinear")
        ref = F.interpolate(x, size=size, mode="bilinear", align_corners=False)
        assert torch.allclose(out, ref, rtol=1e-5, atol=1e-5)
