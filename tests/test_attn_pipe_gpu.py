"""Parity of the software-pipelined attention kernels (csrc/attn_pipe.hip: D = 40, csrc/attn_pipe128.hip: D = 128; round 4) through the C ABI.

Reference op: F.scaled_dot_product_attention without mask (Attention/AttentionMethods.py:107-150, called by CrossAttention.forward
Attention.py:100-124).  The kernel is compared with fp64 / fp32 torch on the same 16-bit inputs and with the kernels it replaces
(attn32*, LDX_ATTN_PIPE=0); the dispatcher reads LDX_ATTN_PIPE / LDX_ATTN_PIPE_MINWG / LDX_ATTN_PIPE_THR once at load; `_Env` re-reads them through ldx_reload_env(), so one process runs all of them.

What is specific to this kernel and therefore tested here:
  * the lazy integer reference maximum: results must not depend on the rescale threshold (THR = 0 / 3 / default agree to rounding: a rescale
    multiplies O, the pending P and the reference by exact powers of two), and score jumps of tens to hundreds of log2 units late in the
    sequence (cdna guide rule 26: the rare branch needs an input that forces it) must come out right — including jumps that would overflow
    16-bit P (> 2^127) if the check came after the exponentials;
  * the reference carried as a hi / lo pair inside the QK^T contraction (|m_ref| >= 256 needs the hi part in bf16);
  * the scale folded into Q (c != 1) versus the engine's calling convention (scale = 1 / log2(e), c == 1);
  * the headline shape (B2 H8 N16384) on a subset of rows, and the shapes the dispatcher must NOT take (fallback stays correct).
Tolerances as tests/test_ops_gpu.py (rel-L2 bf16 4e-3 x 2, fp16 6e-4 x 2): one output rounding plus the 16-bit rounding of P.
"""
import ctypes as C
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

DT = {"bf16": (torch.bfloat16, 0), "f16": (torch.float16, 1)}
TOL = {"bf16": (8e-3, 4e-2), "f16": (1.2e-3, 8e-3)}
LOG2E = 1.4426950408889634


def _p(t):
    return C.c_void_p(t.data_ptr())


def _st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


@pytest.fixture(scope="module")
def L(ldx_lib):
    assert torch.cuda.is_available()
    return ldx_lib


class _Env:
    def __init__(self, **kw):
        self.kw = {k: (None if v is None else str(v)) for k, v in kw.items()}

    @staticmethod
    def _reload():
        # the library reads its dispatch switches once at load (no getenv on the launch path); a test that flips them re-reads them explicitly
        import ldx_amd
        ldx_amd.lib.check(ldx_amd.lib.load().ldx_reload_env(), "ldx_reload_env")

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kw}
        for k, v in self.kw.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        self._reload()

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        self._reload()


def _attn(L, ldx, q, k, v, H, scale, code, pipe, thr=None):
    B, N, Cc = q.shape
    D = Cc // H
    out = torch.full_like(q, float("nan"))
    with _Env(LDX_ATTN_PIPE=pipe, LDX_ATTN_PIPE128=pipe, LDX_ATTN_PIPE_MINWG=1, LDX_ATTN_PIPE_THR=thr):
        ldx.lib.check(L.ldx_op_attention(_p(q), q.stride(1), _p(k), k.stride(1), _p(v), v.stride(1), _p(out), Cc, B, H, N, k.shape[1], D, scale, 0, code, _st()), "attn")
    torch.cuda.synchronize()
    return out


def _ref(q, k, v, H, scale, rows=None, dtype=torch.float64):
    B, N, Cc = q.shape
    D = Cc // H
    qf, kf, vf = (t.to(dtype).reshape(B, -1, H, D).transpose(1, 2) for t in (q, k, v))
    if rows is not None:
        qf = qf[:, :, rows]
    return (torch.softmax(qf @ kf.transpose(-1, -2) * scale, -1) @ vf).transpose(1, 2).reshape(B, -1, Cc)


def _err(got, ref):
    got, ref = got.double(), ref.double()
    return float((got - ref).norm() / ref.norm()), float((got - ref).abs().max() / ref.abs().max())


def _check(got, ref, dt, what):
    assert torch.isfinite(got.float()).all(), f"{what}: non-finite output"
    rel, mx = _err(got, ref)
    r, m = TOL[dt]
    assert rel <= r and mx <= m, f"{what}: rel-L2 {rel:.3e} max {mx:.3e}"
    return rel


def _qkv(B, N, H, D, td, seed, amp=1.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    qkv = (torch.randn(B, N, 3 * H * D, device="cuda", generator=g) * amp).to(td)
    Cc = H * D
    return qkv[..., :Cc], qkv[..., Cc:2 * Cc], qkv[..., 2 * Cc:]


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("B,H,N", [(1, 2, 512), (2, 8, 1024), (1, 8, 4096)])
@pytest.mark.parametrize("conv", ["engine", "plain"])
def test_pipe_vs_fp64_and_attn32(L, ldx, dt, B, H, N, conv):
    td, code = DT[dt]
    D = 40
    q, k, v = _qkv(B, N, H, D, td, seed=B * 1000 + N + H, amp=1.5)
    # "engine": the softmax scale is folded into the q weights and the op is called with 1 / log2(e) (c == 1: Q is used as it is);
    # "plain": scale = 1 / sqrt(D) (c != 1: the kernel rounds Q * c to 16 bit once, which attn32* does not: slightly larger error)
    scale = 1.0 / LOG2E if conv == "engine" else 1.0 / math.sqrt(D)
    if conv == "engine":
        q = (q.float() * (LOG2E / math.sqrt(D))).to(td)
    ref = _ref(q, k, v, H, scale)
    new = _attn(L, ldx, q, k, v, H, scale, code, pipe=1)
    old = _attn(L, ldx, q, k, v, H, scale, code, pipe=0)
    r_new = _check(new, ref, dt, f"pipelined {dt} B{B} H{H} N{N} {conv}")
    r_old = _check(old, ref, dt, f"attn32 {dt} B{B} H{H} N{N} {conv}")
    assert r_new <= 2.5 * r_old + 1e-4, (r_new, r_old)
    if conv == "engine":
        assert r_new <= 1.3 * r_old + 1e-4, (r_new, r_old)       # same roundings as the kernel it replaces


@pytest.mark.parametrize("dt", ["bf16", "f16"])
def test_pipe_threshold_independence(L, ldx, dt):
    """A rescale multiplies O, the pending P and the reference by exact powers of two: forcing it on (almost) every block must not change the result
    beyond the last-bit differences of exp2(s - dl) against exp2(s) * 2^-dl."""
    td, code = DT[dt]
    q, k, v = _qkv(2, 1024, 4, 40, td, seed=77, amp=2.0)
    scale = 1.0 / LOG2E
    base = _attn(L, ldx, q, k, v, 4, scale, code, pipe=1)
    ref = _ref(q, k, v, 4, scale)
    _check(base, ref, dt, "default threshold")
    for thr in (0, 3, -2):
        got = _attn(L, ldx, q, k, v, 4, scale, code, pipe=1, thr=thr)
        _check(got, ref, dt, f"THR={thr}")
        rel, _ = _err(got, base)
        assert rel <= (2e-3 if dt == "bf16" else 3e-4), (thr, rel)


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("jump", [30.0, 90.0, 400.0, 3000.0])
def test_pipe_late_score_jump(L, ldx, dt, jump):
    """Keys late in the sequence whose scores exceed everything before by `jump` log2 units (guide rule 26): 90 is above the bf16 threshold, 400 and
    3000 would overflow P (2^127) if the maximum were checked after the exponentials, 3000 also needs the hi half of the hi / lo reference."""
    td, code = DT[dt]
    B, H, N, D = 1, 2, 1024, 40
    g = torch.Generator(device="cuda").manual_seed(int(jump))
    q = torch.randn(B, N, H * D, device="cuda", generator=g)
    k = torch.randn(B, N, H * D, device="cuda", generator=g)
    v = torch.randn(B, N, H * D, device="cuda", generator=g)
    u = torch.ones(D, device="cuda") / math.sqrt(D)
    amp = math.sqrt(jump) if dt == "bf16" else min(math.sqrt(jump), 40.0)       # fp16 inputs: keep |k| representable
    for h in range(H):
        q[:, :, h * D:(h + 1) * D] = q[:, :, h * D:(h + 1) * D] * 0.5 + u * amp           # every query has a component amp along u
        for j, f in ((333, 0.5), (700, 1.0), (701, 1.0), (1000, 0.8)):                      # keys along u: score ~ amp * f * jump / amp
            k[:, j, h * D:(h + 1) * D] = u * (f * jump / amp)
    q, k, v = q.to(td), k.to(td), v.to(td)
    scale = 1.0 / LOG2E
    ref = _ref(q, k, v, H, scale)
    for thr in (None, 0):
        got = _attn(L, ldx, q, k, v, H, scale, code, pipe=1, thr=thr)
        _check(got, ref, dt, f"jump {jump} thr {thr}")
    old = _attn(L, ldx, q, k, v, H, scale, code, pipe=0)
    _check(old, ref, dt, f"attn32 jump {jump}")


def test_pipe_negative_scores_and_large_reference(L, ldx):
    """All scores far below zero (reference ~ -700: hi / lo pair with a negative hi) and far above (+1500)."""
    td, code = DT["bf16"]
    B, H, N, D = 1, 2, 512, 40
    g = torch.Generator(device="cuda").manual_seed(5)
    for sign in (-1.0, 1.0):
        q = torch.randn(B, N, H * D, device="cuda", generator=g) * 0.3
        k = torch.randn(B, N, H * D, device="cuda", generator=g) * 0.3
        v = torch.randn(B, N, H * D, device="cuda", generator=g)
        q[..., 0::D] = 30.0                  # d = 0 of every head
        k[..., 0::D] = sign * (25.0 if sign < 0 else 50.0)
        q, k, v = q.to(td), k.to(td), v.to(td)
        ref = _ref(q, k, v, H, 1.0 / LOG2E)
        got = _attn(L, ldx, q, k, v, H, 1.0 / LOG2E, code, pipe=1)
        _check(got, ref, "bf16", f"offset scores sign {sign}")


def test_pipe_headline_shape_rows(L, ldx):
    """B2 H8 N16384 D40 (the step's largest launch): 256 query rows against fp32 torch, and the whole output against attn32ap."""
    td, code = DT["bf16"]
    B, H, N, D = 2, 8, 16384, 40
    q, k, v = _qkv(B, N, H, D, td, seed=3)
    q = (q.float() * (LOG2E / math.sqrt(D))).to(td)
    scale = 1.0 / LOG2E
    with _Env(LDX_ATTN_PIPE_MINWG=None):
        new = _attn(L, ldx, q, k, v, H, scale, code, pipe=1)
    old = _attn(L, ldx, q, k, v, H, scale, code, pipe=0)
    rows = torch.arange(0, N, 64, device="cuda")
    ref = _ref(q, k, v, H, scale, rows=rows, dtype=torch.float32)
    _check(new[:, rows], ref, "bf16", "headline rows")
    rel, _ = _err(new, old)
    assert rel <= 8e-3, rel
    again = _attn(L, ldx, q, k, v, H, scale, code, pipe=1)
    assert torch.equal(new, again), "not reproducible"


@pytest.mark.parametrize("B,H,N,M", [(1, 8, 1000, 1024), (1, 8, 1024, 1000), (1, 8, 512, 192), (2, 8, 768, 640)])      # the last one IS taken: Nq != Mk
def test_shapes_the_dispatcher_declines_or_takes_with_other_key_counts(L, ldx, B, H, N, M):
    """Nq % 256 != 0, Mk % 128 != 0, Mk < 256: launch_attention must stay on the older kernels (and stay correct) with the pipelined kernel enabled."""
    td, code = DT["bf16"]
    D = 40
    g = torch.Generator(device="cuda").manual_seed(N + M)
    q = torch.randn(B, N, H * D, device="cuda", generator=g).to(td)
    k = torch.randn(B, M, H * D, device="cuda", generator=g).to(td)
    v = torch.randn(B, M, H * D, device="cuda", generator=g).to(td)
    scale = 1.0 / math.sqrt(D)
    got = _attn(L, ldx, q, k, v, H, scale, code, pipe=1)
    _check(got, _ref(q, k, v, H, scale), "bf16", f"fallback N{N} M{M}")


# ---- D = 128 (attn_pipe128.hip; Flux joint attention, Flux.py:298-348 / 389-418): the same lazy integer reference, but applied in the softmax fma
# (fp32, no pre-scaled Q), row sums from the rounded P on the VALU, 32-key half-slots, MX fp8 output epilogue.

@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("B,H,N", [(1, 2, 512), (1, 24, 1024), (2, 4, 2048)])
def test_pipe128_vs_fp64_and_attn32g(L, ldx, dt, B, H, N):
    td, code = DT[dt]
    D = 128
    q, k, v = _qkv(B, N, H, D, td, seed=B * 1000 + N + H, amp=1.5)
    scale = 1.0 / math.sqrt(D)
    ref = _ref(q, k, v, H, scale)
    new = _attn(L, ldx, q, k, v, H, scale, code, pipe=1)
    old = _attn(L, ldx, q, k, v, H, scale, code, pipe=0)
    r_new = _check(new, ref, dt, f"pipelined D128 {dt} B{B} H{H} N{N}")
    r_old = _check(old, ref, dt, f"attn32g {dt} B{B} H{H} N{N}")
    assert r_new <= 1.3 * r_old + 1e-4, (r_new, r_old)       # same roundings as the kernel it replaces (the denominator sums the rounded P)


@pytest.mark.parametrize("dt", ["bf16", "f16"])
def test_pipe128_threshold_independence(L, ldx, dt):
    td, code = DT[dt]
    q, k, v = _qkv(2, 1024, 4, 128, td, seed=78, amp=2.0)
    scale = 1.0 / math.sqrt(128)
    base = _attn(L, ldx, q, k, v, 4, scale, code, pipe=1)
    ref = _ref(q, k, v, 4, scale)
    _check(base, ref, dt, "default threshold")
    for thr in (0, 3, -2):
        got = _attn(L, ldx, q, k, v, 4, scale, code, pipe=1, thr=thr)
        _check(got, ref, dt, f"THR={thr}")
        rel, _ = _err(got, base)
        assert rel <= (2e-3 if dt == "bf16" else 3e-4), (thr, rel)


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("jump", [30.0, 90.0, 400.0, 3000.0])
def test_pipe128_late_score_jump(L, ldx, dt, jump):
    """Late keys whose scores exceed everything before by `jump` log2 units, in either 32-key half of a block (keys 333 / 700 / 701 / 1000 -> halves
    0, 1, 1, 1 of their blocks; 352 added for a first half): the reference is raised before the exponentials of that half are taken."""
    td, code = DT[dt]
    B, H, N, D = 1, 2, 1024, 128
    g = torch.Generator(device="cuda").manual_seed(int(jump) + 1)
    q = torch.randn(B, N, H * D, device="cuda", generator=g)
    k = torch.randn(B, N, H * D, device="cuda", generator=g)
    v = torch.randn(B, N, H * D, device="cuda", generator=g)
    u = torch.ones(D, device="cuda") / math.sqrt(D)
    amp = math.sqrt(jump) if dt == "bf16" else min(math.sqrt(jump), 40.0)
    for h in range(H):
        q[:, :, h * D:(h + 1) * D] = q[:, :, h * D:(h + 1) * D] * 0.5 + u * amp
        for j, f in ((333, 0.5), (352, 0.7), (700, 1.0), (701, 1.0), (1000, 0.8)):
            k[:, j, h * D:(h + 1) * D] = u * (f * jump / amp)
    q, k, v = q.to(td), k.to(td), v.to(td)
    scale = 1.0 / LOG2E                       # scores in log2 units as constructed
    ref = _ref(q, k, v, H, scale)
    for thr in (None, 0):
        got = _attn(L, ldx, q, k, v, H, scale, code, pipe=1, thr=thr)
        _check(got, ref, dt, f"D128 jump {jump} thr {thr}")
    old = _attn(L, ldx, q, k, v, H, scale, code, pipe=0)
    _check(old, ref, dt, f"attn32g jump {jump}")


def test_pipe128_flux_shape_rows_and_mx_output(L, ldx):
    """B1 H24 N4352 D128 (one Flux launch): 128 query rows against fp32 torch, the whole output against attn32g, reproducible; the MX fp8 output
    (bytes + E8M0 scales, what the Flux fp8 mode consumes) identical to attn32g's epilogue on the same O."""
    td, code = DT["bf16"]
    B, H, N, D = 1, 24, 4352, 128
    Cn = H * D
    q, k, v = _qkv(B, N, H, D, td, seed=4)
    scale = 1.0 / math.sqrt(D)
    with _Env(LDX_ATTN_PIPE_MINWG=None):
        new = _attn(L, ldx, q, k, v, H, scale, code, pipe=1)
    old = _attn(L, ldx, q, k, v, H, scale, code, pipe=0)
    rows = torch.arange(0, N, 34, device="cuda")
    ref = _ref(q, k, v, H, scale, rows=rows, dtype=torch.float32)
    _check(new[:, rows], ref, "bf16", "flux rows")
    rel, _ = _err(new, old)
    assert rel <= 8e-3, rel
    again = _attn(L, ldx, q, k, v, H, scale, code, pipe=1)
    assert torch.equal(new, again), "not reproducible"

    def mx(pipe):
        Y = torch.zeros(B * N, Cn + 16, device="cuda", dtype=torch.uint8)
        S = torch.zeros(Cn // 128, B * N + 9, 4, device="cuda", dtype=torch.uint8)
        with _Env(LDX_ATTN_PIPE128=pipe):
            ldx.lib.check(L.ldx_op_attention_mx(_p(q), q.stride(1), _p(k), k.stride(1), _p(v), v.stride(1), _p(Y), Cn + 16, _p(S), B * N + 9, B, H, N, N, scale, code, _st()), "attention_mx")
        torch.cuda.synchronize()
        return Y.cpu()[:, :Cn].clone(), S.cpu()[:, :B * N].clone()
    y1, s1 = mx(1)
    y0, s0 = mx(0)
    # the two kernels' O differ in the last bit here and there (different summation order of l): compare the dequantised values
    def deq(y, s):
        e = s[..., :4].reshape(Cn // 128, B * N, 4).permute(1, 0, 2).reshape(B * N, Cn // 32).to(torch.int32)      # one E8M0 byte per 32 columns
        f = y.view(torch.float8_e4m3fn).float().reshape(B * N, Cn // 32, 32)
        return (f * torch.exp2(e.float() - 127.0)[..., None]).reshape(B * N, Cn)
    d1, d0 = deq(y1, s1), deq(y0, s0)
    relq = float((d1 - d0).norm() / d0.norm())
    assert relq <= 3e-2, relq                   # fp8 (e4m3, 3 mantissa bits) of two O's that agree to 8e-3
    relo = float((d1 - new.reshape(B * N, Cn).float().cpu()).norm() / new.float().norm())
    assert relo <= 4e-2, relo                   # and it IS the quantised O of this kernel


@pytest.mark.parametrize("B,H,N,M", [(1, 8, 1100, 1100), (1, 8, 1024, 1000), (1, 8, 512, 192), (2, 8, 768, 640)])      # the last one IS taken: Nq != Mk
def test_pipe128_declined_shapes_stay_correct(L, ldx, B, H, N, M):
    td, code = DT["bf16"]
    D = 128
    g = torch.Generator(device="cuda").manual_seed(N + M)
    q = torch.randn(B, N, H * D, device="cuda", generator=g).to(td)
    k = torch.randn(B, M, H * D, device="cuda", generator=g).to(td)
    v = torch.randn(B, M, H * D, device="cuda", generator=g).to(td)
    scale = 1.0 / math.sqrt(D)
    got = _attn(L, ldx, q, k, v, H, scale, code, pipe=1)
    _check(got, _ref(q, k, v, H, scale), "bf16", f"D128 N{N} M{M}")
