"""MX fp8 (OCP microscaling) operands on a real MI355X, through the C ABI.

ldx_op_mx_quant is checked bit-for-bit (element bytes and E8M0 scale bytes) against a torch restatement of the stated
rule: blocks of 32 consecutive k, scale = 2^ceil(log2(amax / 448)), element = e4m3fn(x / scale) rounded to nearest even.
ldx_op_gemm_mx is checked against an fp64 matmul of the DEQUANTISED operands: products of two e4m3 values and power-of-two
scales are exact, so only the fp32 accumulation order differs (rel-L2 <= 5e-5 on the fp32 output after the tanh-GELU epilogue; the 16-bit output adds
one rounding).
"""
import ctypes as C
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DT = {"bf16": (torch.bfloat16, 0), "f16": (torch.float16, 1)}


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def mx_quant_ref(x):
    """x: fp32 [rows][K] on the CPU -> (e4m3 bytes uint8 [rows][K], E8M0 bytes uint8 [rows][K/32], dequantised fp32)."""
    rows, K = x.shape
    xb = x.view(rows, K // 32, 32)
    amax = xb.abs().amax(-1)
    r = amax * torch.tensor(np.float32(1.0) / np.float32(448.0))
    bits = r.view(torch.int32)
    e = ((bits >> 23) & 0xFF) + ((bits & 0x7FFFFF) != 0).to(torch.int32)
    e = e.clamp(1, 253)
    inv = ((254 - e) << 23).view(torch.float32)
    scale = (e << 23).view(torch.float32)
    q = (xb * inv[..., None]).to(torch.float8_e4m3fn)
    deq = (q.float() * scale[..., None]).view(rows, K)
    return q.view(torch.uint8).view(rows, K), e.to(torch.uint8), deq


def scales_layout(e, ld):
    """[rows][K/32] bytes -> the K-tile-major dword layout [K/128][ld] (as uint8 [K/128][ld][4])."""
    rows, nkb = e.shape
    out = torch.zeros(nkb // 4, ld, 4, dtype=torch.uint8)
    out[:, :rows, :] = e.view(rows, nkb // 4, 4).permute(1, 0, 2)
    return out


def quant_gpu(L, ldx, X16, K, code, ldy=None, s_ld=None):
    rows = X16.shape[0]
    ldy = ldy or K
    s_ld = s_ld or rows
    Y = torch.zeros(rows, ldy, device="cuda", dtype=torch.uint8)
    S = torch.zeros(K // 128, s_ld, 4, device="cuda", dtype=torch.uint8)
    ldx.lib.check(L.ldx_op_mx_quant(_p(X16), X16.stride(0), rows, K, _p(Y), ldy, _p(S), s_ld, code, _st()), "mx_quant")
    return Y, S


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("rows,K,pad", [(300, 384, 0), (77, 128, 8), (1000, 3072, 64)])
def test_mx_quant_bit_exact(ldx_lib, ldx, dt, rows, K, pad):
    L = ldx_lib
    td, code = DT[dt]
    g = torch.Generator().manual_seed(rows + K)
    x = torch.randn(rows, K + pad, generator=g) * torch.exp(2.0 * torch.randn(rows, 1, generator=g))
    x[3, 32:64] = 0.0                                     # an all-zero block
    x[5, 0] = 448.0 * 4                                   # amax / 448 exactly a power of two
    x[7, 64:96] *= 1e-30 if dt == "bf16" else 1e-6        # tiny block (sub-normal quotient for bf16)
    X16 = x.to(td).cuda()
    Y, S = quant_gpu(L, ldx, X16, K, code, ldy=K + 16, s_ld=rows + 5)
    q, e, _ = mx_quant_ref(X16.cpu().float()[:, :K].contiguous())
    assert torch.equal(S.cpu()[:, :rows, :], scales_layout(e, rows)), "E8M0 scales differ"
    got = Y.cpu()[:, :K]
    # -0 and +0 are the same value: compare with the sign of zeros cleared
    gz, qz = got.clone(), q.clone()
    gz[(gz & 0x7F) == 0] = 0
    qz[(qz & 0x7F) == 0] = 0
    assert torch.equal(gz, qz), f"{int((gz != qz).sum())} element bytes differ"


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("M,N,K", [(300, 200, 384), (1024, 640, 3072), (256, 3072, 3072), (130, 64, 128), (4096, 320, 1024)])
def test_gemm_mx(ldx_lib, ldx, dt, M, N, K):
    L = ldx_lib
    td, code = DT[dt]
    g = torch.Generator().manual_seed(M + N + K)
    A = (torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, 1, generator=g))).to(td)
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(td)
    bias = torch.randn(N, generator=g)
    R = torch.randn(M, N + 8, generator=g).to(td)
    A8, SA = quant_gpu(L, ldx, A.cuda(), K, code, s_ld=M + 3)
    W8, SW = quant_gpu(L, ldx, W.cuda(), K, code)
    Cc = torch.zeros(M, N + 8, device="cuda", dtype=td)
    Cf = torch.zeros(M, N, device="cuda", dtype=torch.float32)
    Rg = R.cuda()
    ldx.lib.check(L.ldx_op_gemm_mx(_p(A8), K, _p(SA), M + 3, _p(W8), _p(SW), N, M, N, K, _p(bias.cuda()), 2, _p(Rg), N + 8,
                                   _p(Cc), N + 8, _p(Cf), N, None, 0, None, 0, code, _st()), "gemm_mx")
    torch.cuda.synchronize()
    _, _, Ad = mx_quant_ref(A.float())
    _, _, Wd = mx_quant_ref(W.float())
    ref = (Ad.double() @ Wd.double().t()) + bias.double()
    ref = torch.nn.functional.gelu(ref, approximate="tanh") + R[:, :N].double()
    got = Cf.cpu().double()
    rel = float((got - ref).norm() / ref.norm())
    assert math.isfinite(rel) and rel <= 5e-5, f"fp32 output rel-L2 {rel:.3e}"
    got16 = Cc.cpu()[:, :N].double()
    rel16 = float((got16 - ref).norm() / ref.norm())
    assert rel16 <= (4e-3 if dt == "bf16" else 6e-4), f"16-bit output rel-L2 {rel16:.3e}"
    # and the quantisation itself: MX fp8 operands vs the 16-bit operands (documented accuracy class of the mode)
    full = A.double() @ W.double().t()
    qerr = float(((Ad.double() @ Wd.double().t()) - full).norm() / full.norm())
    assert qerr < 6e-2, f"quantisation error {qerr:.3e}"


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("M,N,K", [(300, 256, 384), (1024, 1280, 3072), (4096, 1024, 256), (77, 128, 128)])
def test_gemm_mx_quantised_output(ldx_lib, ldx, dt, M, N, K):
    """MX output epilogue == 16-bit output followed by ldx_op_mx_quant, bit for bit (bytes and scales)."""
    L = ldx_lib
    td, code = DT[dt]
    g = torch.Generator().manual_seed(M * 3 + N + K)
    A = torch.randn(M, K, generator=g).to(td).cuda()
    W = (torch.randn(N, K, generator=g) / math.sqrt(K) * torch.exp(torch.randn(N, 1, generator=g))).to(td).cuda()
    bias = torch.randn(N, generator=g).cuda()
    A8, SA = quant_gpu(L, ldx, A, K, code)
    W8, SW = quant_gpu(L, ldx, W, K, code)
    Cc = torch.zeros(M, N, device="cuda", dtype=td)
    ldx.lib.check(L.ldx_op_gemm_mx(_p(A8), K, _p(SA), M, _p(W8), _p(SW), N, M, N, K, _p(bias), 2, None, 0, _p(Cc), N, None, 0,
                                   None, 0, None, 0, code, _st()), "gemm_mx")
    Y2, S2 = quant_gpu(L, ldx, Cc, N, code, ldy=N + 32, s_ld=M + 7)
    Y1 = torch.zeros(M, N + 32, device="cuda", dtype=torch.uint8)
    S1 = torch.zeros(N // 128, M + 7, 4, device="cuda", dtype=torch.uint8)
    ldx.lib.check(L.ldx_op_gemm_mx(_p(A8), K, _p(SA), M, _p(W8), _p(SW), N, M, N, K, _p(bias), 2, None, 0, None, 0, None, 0,
                                   _p(Y1), N + 32, _p(S1), M + 7, code, _st()), "gemm_mx C8")
    torch.cuda.synchronize()
    assert torch.equal(S1.cpu(), S2.cpu()), "scales differ"
    a, b = Y1.cpu()[:, :N].clone(), Y2.cpu()[:, :N].clone()
    a[(a & 0x7F) == 0] = 0
    b[(b & 0x7F) == 0] = 0
    assert torch.equal(a, b), f"{int((a != b).sum())} bytes differ"


def test_mx_bad_args(ldx_lib, ldx):
    L = ldx_lib
    x = torch.zeros(4, 128, device="cuda", dtype=torch.bfloat16)
    y = torch.zeros(4, 128, device="cuda", dtype=torch.uint8)
    s = torch.zeros(1, 4, 4, device="cuda", dtype=torch.uint8)
    assert L.ldx_op_mx_quant(_p(x), 128, 4, 96, _p(y), 128, _p(s), 4, 0, _st()) != 0          # K % 128
    assert L.ldx_op_mx_quant(_p(x), 128, 4, 128, _p(y), 128, _p(s), 2, 0, _st()) != 0         # scales_ld < rows
    assert L.ldx_op_gemm_mx(_p(y), 128, _p(s), 4, _p(y), _p(s), 4, 4, 4, 64, None, 0, None, 0, _p(x), 128, None, 0, None, 0, None, 0, 0, _st()) != 0


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("B,H,N", [(1, 24, 4352), (2, 4, 1100)])
def test_attention_mx_output(ldx_lib, ldx, dt, B, H, N):
    """Attention with the MX output epilogue == ldx_op_attention followed by ldx_op_mx_quant, bit for bit."""
    L = ldx_lib
    td, code = DT[dt]
    D, Cn = 128, H * 128
    g = torch.Generator().manual_seed(B + H + N)
    qkv = torch.randn(B, N, 3 * Cn, generator=g).to(td).cuda()
    O = torch.zeros(B * N, Cn, device="cuda", dtype=td)
    sc = 1.0 / math.sqrt(D)
    q, k, v = qkv[..., :Cn], qkv[..., Cn:2 * Cn], qkv[..., 2 * Cn:]
    ldx.lib.check(L.ldx_op_attention(_p(q), 3 * Cn, _p(k), 3 * Cn, _p(v), 3 * Cn, _p(O), Cn, B, H, N, N, D, sc, 0, code, _st()), "attention")
    Y2, S2 = quant_gpu(L, ldx, O, Cn, code, ldy=Cn + 16, s_ld=B * N + 9)
    Y1 = torch.zeros(B * N, Cn + 16, device="cuda", dtype=torch.uint8)
    S1 = torch.zeros(Cn // 128, B * N + 9, 4, device="cuda", dtype=torch.uint8)
    ldx.lib.check(L.ldx_op_attention_mx(_p(q), 3 * Cn, _p(k), 3 * Cn, _p(v), 3 * Cn, _p(Y1), Cn + 16, _p(S1), B * N + 9, B, H, N, N, sc, code, _st()),
                  "attention_mx")
    torch.cuda.synchronize()
    assert torch.equal(S1.cpu()[:, :B * N], S2.cpu()[:, :B * N]), "scales differ"
    a, b = Y1.cpu()[:, :Cn].clone(), Y2.cpu()[:, :Cn].clone()
    a[(a & 0x7F) == 0] = 0
    b[(b & 0x7F) == 0] = 0
    # bf16: identical.  f16: hipcc rounds o / l to f16 in one step (v_fma_mixlo_f16) in one kernel and in two (fp32 product, then
    # v_cvt_f16_f32) in the other; the rare values this moves across an e4m3 tie differ by ONE fp8 code (measured 3 of 13.4 M)
    nd = int((a != b).sum())
    assert nd <= (0 if dt == "bf16" else max(1, a.numel() // 1_000_000)), f"{nd} bytes differ"
    assert int((a.to(torch.int16) - b.to(torch.int16)).abs().max()) <= 1
    # too small a grid for the kernel that implements the epilogue: refused, not silently different
    assert L.ldx_op_attention_mx(_p(q), 3 * Cn, _p(k), 3 * Cn, _p(v), 3 * Cn, _p(Y1), Cn + 16, _p(S1), B * N + 9, 1, 1, 100, 100, sc, code, _st()) != 0


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("rows,Cn", [(1000, 3072), (77, 256), (333, 1280), (64, 4096)])
def test_layernorm_mx_output(ldx_lib, ldx, dt, rows, Cn):
    """LayerNorm with the MX output epilogue == ldx_op_layernorm followed by ldx_op_mx_quant."""
    L = ldx_lib
    td, code = DT[dt]
    g = torch.Generator().manual_seed(rows + Cn)
    X = (torch.randn(rows, Cn, generator=g) * 3 + 0.5).to(td).cuda()
    gm, bt = torch.randn(Cn, generator=g).cuda(), torch.randn(Cn, generator=g).cuda()
    Y = torch.zeros(rows, Cn, device="cuda", dtype=td)
    ldx.lib.check(L.ldx_op_layernorm(_p(X), Cn, _p(Y), Cn, rows, Cn, 1e-6, _p(gm), _p(bt), code, _st()), "ln")
    Y2, S2 = quant_gpu(L, ldx, Y, Cn, code, ldy=Cn + 16, s_ld=rows + 3)
    Y1 = torch.zeros(rows, Cn + 16, device="cuda", dtype=torch.uint8)
    S1 = torch.zeros(Cn // 128, rows + 3, 4, device="cuda", dtype=torch.uint8)
    ldx.lib.check(L.ldx_op_layernorm_mx(_p(X), Cn, rows, Cn, 1e-6, _p(gm), _p(bt), _p(Y1), Cn + 16, _p(S1), rows + 3, code, _st()), "ln_mx")
    torch.cuda.synchronize()
    assert torch.equal(S1.cpu(), S2.cpu()), "scales differ"
    a, b = Y1.cpu()[:, :Cn].clone(), Y2.cpu()[:, :Cn].clone()
    a[(a & 0x7F) == 0] = 0
    b[(b & 0x7F) == 0] = 0
    nd = int((a != b).sum())
    assert nd <= (0 if dt == "bf16" else max(1, a.numel() // 1_000_000)), f"{nd} bytes differ"
