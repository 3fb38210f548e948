"""The 256-row ping-pong GEMM / implicit-GEMM main loop (csrc/gemm_pp.inc) on the op tests' shapes.

The cost model only picks that kernel for large problems, which the op tests (odd sizes, ragged M / N / K, stride 2, nearest-resize,
fused skip segment, split-K ranges that start inside the second K segment, GEGLU, MX fp8 operands and quantised output) are not.  The
tile is forced with LDX_GEMM_TILE (read once per process), so each tile width runs the GEMM / conv / MX op tests in a subprocess: same
tests, same torch-fp32 / fp64 references and tolerances, different kernel underneath (LDS-DMA zero-fill of out-of-range rows included)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("tile", ["256128", "256160", "256192", "256224", "256256"])   # 256192 / 256224: plain GEMMs (16-bit and MX fp8); convs and GEGLU fall to 256128
def test_op_tests_on_forced_pingpong_tiles(ldx_lib, tile):
    env = dict(os.environ, LDX_GEMM_TILE=tile)
    sel = "test_gemm or test_conv3x3 or test_gemm_mx"
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", "tests/test_ops_gpu.py", "tests/test_mx_gpu.py", "-k", sel],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    tail = r.stdout[-1500:]
    print(tail, r.stderr[-500:])
    assert r.returncode == 0 and " passed" in tail and "failed" not in tail, tail


def test_op_tests_on_the_forced_ring_tile(ldx_lib):
    """Round 5: the 64 x 160 LDS-DMA ring kernel (csrc/gemm_ring.hip; plain GEMMs of the 32^2 level: 2048 x 1280 x 1280 in one round of 256 workgroups,
    three to four K-tiles in flight) forced onto the GEMM op tests — ragged M / N, bias / residual / rowvec / fp32-output epilogues, GroupNorm statistics;
    split-K, ragged K, convs and GEGLU fall back to their own tiles."""
    env = dict(os.environ, LDX_GEMM_TILE="64160")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", "tests/test_ops_gpu.py", "tests/test_rowgemm_gpu.py", "-k", "test_gemm or rowgemm"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    tail = r.stdout[-1500:]
    print(tail, r.stderr[-500:])
    assert r.returncode == 0 and " passed" in tail and "failed" not in tail, tail


@pytest.mark.parametrize("shape", [(2048, 1280, 1280), (2048, 3840, 1280), (2000, 1280, 640), (8192, 640, 2560)])
def test_ring_gemm_shapes_match_torch(ldx_lib, shape):
    """Un-forced: shapes the planner sends to the ring kernel (M 2048, N = K = 1280: the SD1.5 32^2 level's proj_in / to_out / to_q / proj_out) and, with
    LDX_GEMM_TILE unset, whatever it picks for the others — bias + residual epilogue against torch fp32 on the same bf16 operands, bit-identical repeats,
    K-tile counts below / at / above the ring depth."""
    import ctypes as C
    import torch
    L = ldx_lib
    p = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    M, N, K = shape
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A = torch.randn(M, K, device="cuda", generator=g).bfloat16(); W = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
    bias = torch.randn(N, device="cuda", generator=g); R = torch.randn(M, N, device="cuda", generator=g).bfloat16()
    outs = []
    for _ in range(2):
        Cc = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
        assert L.ldx_op_gemm(p(A), K, p(W), M, N, K, p(bias), None, 0, 1, 0, p(R), N, p(Cc), N, None, 0, 0, st) == 0
        outs.append(Cc)
    torch.cuda.synchronize()
    ref = A.float() @ W.float().t() + bias + R.float()
    rel = float((outs[0].float() - ref).norm() / ref.norm())
    print(f"gemm {shape}: rel-L2 {rel:.3e}")
    assert rel < 4e-3
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("K", [128, 192, 256, 320, 384])
def test_ring_gemm_short_k(ldx_lib, K):
    """K-tile counts 2 .. 6 around the ring depth of five (prologue / tail vmcnt cases), forced onto the ring tile through LDX_GEMM_TILE in a subprocess."""
    code = (f"import sys, ctypes as C, torch; sys.path.insert(0, {ROOT!r}); import ldx_amd as ldx; L = ldx.lib.load()\n"
            "p = lambda t: C.c_void_p(t.data_ptr()); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)\n"
            f"M, N, K = 1000, 480, {K}\n"
            "g = torch.Generator(device='cuda').manual_seed(K)\n"
            "A = torch.randn(M, K, device='cuda', generator=g).bfloat16(); W = (torch.randn(N, K, device='cuda', generator=g) / K ** 0.5).bfloat16()\n"
            "Cc = torch.full((M, N), float('nan'), device='cuda', dtype=torch.bfloat16)\n"
            "assert L.ldx_op_gemm(p(A), K, p(W), M, N, K, None, None, 0, 1, 0, None, 0, p(Cc), N, None, 0, 0, st) == 0\n"
            "ref = A.float() @ W.float().t(); rel = float((Cc.float() - ref).norm() / ref.norm()); print('REL', rel); assert rel < 4e-3\n")
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, LDX_GEMM_TILE="64160"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "REL" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])


def test_large_shapes_pick_the_pingpong_kernel_and_match_torch(ldx, ldx_lib):
    """Un-forced: shapes the cost model sends to the ping-pong kernel (plain GEMM with long K, 3x3 conv at the level-0 size), against
    torch fp32 on the same 16-bit operands."""
    import ctypes as C
    import torch
    L = ldx_lib
    p = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device="cuda").manual_seed(5)
    M, N, K = 4096, 1280, 2048
    A = torch.randn(M, K, device="cuda", generator=g).bfloat16(); W = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
    bias = torch.randn(N, device="cuda", generator=g); R = torch.randn(M, N, device="cuda", generator=g).bfloat16()
    Cc = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    assert L.ldx_op_gemm(p(A), K, p(W), M, N, K, p(bias), None, 0, 1, 0, p(R), N, p(Cc), N, None, 0, 0, st) == 0
    ref = A.float() @ W.float().t() + bias + R.float()
    assert float((Cc.float() - ref).norm() / ref.norm()) < 4e-3
    B, H, Cin, Cout = 2, 64, 320, 320
    X = torch.randn(B, H, H, Cin, device="cuda", generator=g).bfloat16()
    Wc = (torch.randn(Cout, Cin, 3, 3, device="cuda", generator=g) / (9 * Cin) ** 0.5)
    Wp = Wc.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous().bfloat16()
    Y = torch.empty(B * H * H, Cout, device="cuda", dtype=torch.bfloat16)
    assert L.ldx_op_conv3x3(p(X), Cin, p(Wp), B, H, H, Cin, Cout, 1, H, H, 0, None, None, 0, None, 0, p(Y), Cout, 0, st) == 0
    ref = torch.nn.functional.conv2d(X.float().permute(0, 3, 1, 2), Wp.float().reshape(Cout, 3, 3, Cin).permute(0, 3, 1, 2), padding=1)
    ref = ref.permute(0, 2, 3, 1).reshape(B * H * H, Cout)
    assert float((Y.float() - ref).norm() / ref.norm()) < 4e-3


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("case", ["gemm16", "gemm2_16", "gemm_mx", "gemm2_mx"])
def test_multi_round_and_two_problem_launches(ldx_lib, case, dt):
    """Launches of more than one round of 256-row tiles (one or two problems per launch, 16-bit or MX fp8 operands, ragged edges, quantised
    output on the 256 x 192 tile): against torch fp32 / the fp64 product of the dequantised operands."""
    import torch
    import _pp_multiround_cases as cases
    td, code = {"bf16": (torch.bfloat16, 0), "f16": (torch.float16, 1)}[dt]
    outs, refs = cases.CASES[case](ldx_lib, td, code)
    torch.cuda.synchronize()
    for o, r in zip(outs, refs):
        if o.dtype == torch.uint8:
            assert torch.equal(o.cpu(), r.cpu()), "quantised output differs from quantising the 16-bit output"
            continue
        rel = float((o.double() - r.double()).norm() / r.double().norm())
        tol = 5e-5 if o.dtype == torch.float32 else (4e-3 if dt == "bf16" else 6e-4)
        assert rel <= tol, f"{case} {o.dtype}: rel-L2 {rel:.3e}"


@pytest.mark.parametrize("env", [{"LDX_ATTN32_AP": "0"},                                   # the 4-wave attn32_kernel (taken by nothing once the 8-wave kernel is on)
                                 {"LDX_ATTN32_AP": "2"},                                   # strict phases
                                 {"LDX_ATTN32_AP": "1", "LDX_ATTN32_AP_MINWG": "1"}])      # the 8-wave kernel on every D = 40 grid incl. one ragged 512-query block
def test_attention_tests_on_forced_d40_kernels(ldx_lib, env):
    """The D = 40 attention has three kernels behind one dispatcher (attention.hip launch_attn32); the environment switches are read once per
    process, so the attention op tests (same torch-fp32 / fp64 references and tolerances) run in a subprocess per variant."""
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", "tests/test_ops_gpu.py", "-k", "test_attention"],
                       cwd=ROOT, env=dict(os.environ, **env), capture_output=True, text=True, timeout=1500)
    tail = r.stdout[-1500:]
    print(tail, r.stderr[-500:])
    assert r.returncode == 0 and " passed" in tail and "failed" not in tail, tail
