"""Multi-round ping-pong GEMM cases of tests/test_pingpong_gpu.py (not a test module): each returns (outputs, references) of launches
whose 256-row tiles make more than one round of the 256 CUs.  Run as a script it prints one sha256 per case and dtype (used to A/B
kernel variants that must keep the arithmetic order, e.g. the persistent-tile experiment of profiles/ubench/README.md)."""
import ctypes as C
import hashlib
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _quant(L, X16, K, code, s_ld):
    rows = X16.shape[0]
    Y = torch.zeros(rows, K, device="cuda", dtype=torch.uint8)
    S = torch.zeros(K // 128, s_ld, 4, device="cuda", dtype=torch.uint8)
    assert L.ldx_op_mx_quant(_p(X16), X16.stride(0), rows, K, _p(Y), K, _p(S), s_ld, code, _st()) == 0
    return Y, S


def case_gemm16(L, td, code):
    """ragged M, N and K; bias + residual; the cost model picks 256 x 192 tiles: 33 x 14 = 462 of them"""
    M, N, K = 8200, 2576, 1096
    g = torch.Generator(device="cuda").manual_seed(11)
    A = torch.randn(M, K, device="cuda", generator=g).to(td)
    W = (torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K)).to(td)
    bias = torch.randn(N, device="cuda", generator=g)
    R = torch.randn(M, N, device="cuda", generator=g).to(td)
    Cc = torch.zeros(M, N, device="cuda", dtype=td)
    assert L.ldx_op_gemm(_p(A), K, _p(W), M, N, K, _p(bias), None, 0, 1, 0, _p(R), N, _p(Cc), N, None, 0, code, _st()) == 0
    ref = A.float() @ W.float().t() + bias + R.float()
    return [Cc], [ref]


def case_gemm2_16(L, td, code):
    """two problems (4096 and 520 rows, different N and K) in one launch: 256 x 192 tiles, 16 x 24 + 3 x 20 = 444"""
    g = torch.Generator(device="cuda").manual_seed(12)
    outs, refs, args = [], [], []
    for (M, N, K) in ((4096, 4608, 1024), (520, 3800, 1160)):
        A = torch.randn(M, K, device="cuda", generator=g).to(td)
        W = (torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K)).to(td)
        bias = torch.randn(N, device="cuda", generator=g)
        Cc = torch.zeros(M, N, device="cuda", dtype=td)
        args += [_p(A), K, _p(W), M, N, K, _p(bias), _p(Cc), N]
        outs.append(Cc); refs.append(A.float() @ W.float().t() + bias)
        args_keep.append((A, W, bias))
    assert L.ldx_op_gemm2(*args, code, _st()) == 0
    return outs, refs


args_keep = []


def _deq(Y, S, rows, K):
    e = S[:, :rows, :].permute(1, 0, 2).reshape(rows, K // 32).to(torch.int32)
    scale = (e << 23).view(torch.float32)
    return (Y.view(torch.float8_e4m3fn).float().view(rows, K // 32, 32) * scale[..., None]).view(rows, K)


def case_gemm_mx(L, td, code):
    """MX fp8 operands, ragged M, 16-bit and fp32 outputs, tanh-GELU epilogue: 256 x 224 / 192 tiles, several rounds"""
    M, N, K = 8200, 2560, 2048
    g = torch.Generator(device="cuda").manual_seed(13)
    A = (torch.randn(M, K, device="cuda", generator=g) * torch.exp(torch.randn(M, 1, device="cuda", generator=g))).to(td)
    W = (torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K)).to(td)
    bias = torch.randn(N, device="cuda", generator=g)
    A8, SA = _quant(L, A, K, code, M + 3)
    W8, SW = _quant(L, W, K, code, N)
    Cc = torch.zeros(M, N, device="cuda", dtype=td)
    Cf = torch.zeros(M, N, device="cuda", dtype=torch.float32)
    assert L.ldx_op_gemm_mx(_p(A8), K, _p(SA), M + 3, _p(W8), _p(SW), N, M, N, K, _p(bias), 2, None, 0, _p(Cc), N, _p(Cf), N,
                            None, 0, None, 0, code, _st()) == 0
    ref = torch.nn.functional.gelu((_deq(A8, SA, M, K).double() @ _deq(W8, SW, N, K).double().t()) + bias.double(), approximate="tanh")
    # quantised output (256 x 192 tiles): against the 16-bit output quantised by the stand-alone kernel
    Y1 = torch.zeros(M, N, device="cuda", dtype=torch.uint8)
    S1 = torch.zeros(N // 128, M, 4, device="cuda", dtype=torch.uint8)
    assert L.ldx_op_gemm_mx(_p(A8), K, _p(SA), M + 3, _p(W8), _p(SW), N, M, N, K, _p(bias), 2, None, 0, None, 0, None, 0,
                            _p(Y1), N, _p(S1), M, code, _st()) == 0
    Y2, S2 = _quant(L, Cc, N, code, M)
    Y1 = Y1.clone(); Y2 = Y2.clone()
    Y1[(Y1 & 0x7F) == 0] = 0
    Y2[(Y2 & 0x7F) == 0] = 0
    return [Cf, Cc, Y1, S1], [ref, ref, Y2, S2]


def case_gemm2_mx(L, td, code):
    g = torch.Generator(device="cuda").manual_seed(14)
    outs, refs, args = [], [], []
    for (M, N, K) in ((4096, 4608, 2048), (520, 3840, 2304)):
        A = torch.randn(M, K, device="cuda", generator=g).to(td)
        W = (torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K)).to(td)
        bias = torch.randn(N, device="cuda", generator=g)
        A8, SA = _quant(L, A, K, code, M)
        W8, SW = _quant(L, W, K, code, N)
        Cc = torch.zeros(M, N, device="cuda", dtype=td)
        args += [_p(A8), K, _p(SA), M, _p(W8), _p(SW), N, M, N, K, _p(bias), _p(Cc), N]
        outs.append(Cc); refs.append((_deq(A8, SA, M, K).double() @ _deq(W8, SW, N, K).double().t()) + bias.double())
        args_keep.append((A8, SA, W8, SW, bias))
    assert L.ldx_op_gemm2_mx(*args, code, _st()) == 0
    return outs, refs


CASES = {"gemm16": case_gemm16, "gemm2_16": case_gemm2_16, "gemm_mx": case_gemm_mx, "gemm2_mx": case_gemm2_mx}

if __name__ == "__main__":
    import ldx_amd as ldx
    L = ldx.lib.load()
    for name, fn in CASES.items():
        for td, code in ((torch.bfloat16, 0), (torch.float16, 1)):
            outs, _ = fn(L, td, code)
            torch.cuda.synchronize()
            h = hashlib.sha256()
            for o in outs:
                h.update(o.cpu().contiguous().view(torch.uint8).numpy().tobytes())
            print(name, code, h.hexdigest())
