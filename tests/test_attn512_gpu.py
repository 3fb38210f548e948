"""Flash attention for ONE head of D = 512 (csrc/attn512.hip, round 5): the VAE mid-block AttnBlock (Attention.py:127-178 -> AttentionMethods.py:175-197,
1 head, D = C = 512, N = h w) without the N x N scores going through HBM.  Through the C ABI (ldx_op_attention) against fp64 torch on the same 16-bit
operands: square and rectangular shapes, ragged N (rows past Mk land as zeros by LDS-DMA and are masked), the key-split path (few query blocks ->
up to 8 workgroups per query block + merge launch), several heads / batches, a late dominant key (online-softmax rescale incl. the asm rescale of the
accumulator file), fused q|k|v row strides as the engine uses them; and the engine-level check: the full-size VAE decode with the flash kernel against the
GEMM -> softmax -> GEMM path it replaces (LDX_ATTN512=0, subprocess)."""
import ctypes as C
import hashlib
import math
import os
import subprocess
import sys
import textwrap

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DT = {"bf16": (torch.bfloat16, 0), "f16": (torch.float16, 1)}


def _p(t):
    return C.c_void_p(t.data_ptr())


def _st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _run(L, ldx, q, k, v, H, scale, code, ldq=None, ldk=None, ldv=None):
    B, Nq, Cc = q.shape
    out = torch.full((B, Nq, Cc), float("nan"), device="cuda", dtype=q.dtype)
    ldx.lib.check(L.ldx_op_attention(_p(q), ldq or q.stride(1), _p(k), ldk or k.stride(1), _p(v), ldv or v.stride(1), _p(out), Cc, B, H, Nq, k.shape[1], Cc // H,
                                     scale, 0, code, _st()), "attn512")
    torch.cuda.synchronize()
    return out


def _ref(q, k, v, H, scale):
    B, Nq, Cc = q.shape
    D = Cc // H
    qf, kf, vf = (t.double().reshape(B, -1, H, D).transpose(1, 2) for t in (q, k, v))
    return (torch.softmax(qf @ kf.transpose(-1, -2) * scale, -1) @ vf).transpose(1, 2).reshape(B, Nq, Cc)


def _rel(a, b):
    return float((a.double() - b).norm() / b.norm())


# (B, H, Nq, Mk): 128 x 128 one block; ragged both ways; splits = 8 (2 query blocks x 4096 keys); rectangular; batch x heads; 16384 keys = the 1024^2 VAE shape (2 splits)
CASES = [(1, 1, 128, 128), (1, 1, 200, 216), (1, 1, 256, 4096), (1, 1, 1000, 333), (2, 2, 384, 1024), (1, 1, 4096, 4096)]


@pytest.mark.parametrize("dt,tol", [("bf16", 4e-3), ("f16", 6e-4)])
@pytest.mark.parametrize("case", CASES)
def test_attn512_vs_fp64(ldx, ldx_lib, dt, tol, case):
    td, code = DT[dt]
    B, H, Nq, Mk = case
    g = torch.Generator(device="cuda").manual_seed(sum(case))
    q = torch.randn(B, Nq, H * 512, device="cuda", generator=g).to(td)
    k = torch.randn(B, Mk, H * 512, device="cuda", generator=g).to(td)
    v = torch.randn(B, Mk, H * 512, device="cuda", generator=g).to(td)
    scale = 1.0 / math.sqrt(512)
    out = _run(ldx_lib, ldx, q, k, v, H, scale, code)
    assert torch.isfinite(out.float()).all()
    r = _rel(out, _ref(q, k, v, H, scale))
    print(f"attn512 {case} {dt}: rel-L2 {r:.3e}")
    assert r <= tol
    assert torch.equal(out, _run(ldx_lib, ldx, q, k, v, H, scale, code))          # deterministic (fixed split order in the merge)


def test_attn512_full_vae_shape_and_fused_qkv_strides(ldx, ldx_lib):
    """N = 16 384, q | k | v as column ranges of one [N][1536] buffer (the engine's fused projection), checked on 512 query rows against fp64."""
    td, code = DT["bf16"]
    N = 16384
    g = torch.Generator(device="cuda").manual_seed(4)
    qkv = torch.randn(1, N, 1536, device="cuda", generator=g).to(td)
    q, k, v = qkv[..., :512], qkv[..., 512:1024], qkv[..., 1024:]
    scale = 1.0 / math.sqrt(512)
    out = _run(ldx_lib, ldx, q, k, v, 1, scale, code, 1536, 1536, 1536)
    rows = torch.arange(0, N, 32, device="cuda")
    ref = torch.softmax(q[0, rows].double() @ k[0].double().t() * scale, -1) @ v[0].double()
    r = _rel(out[0, rows], ref)
    print(f"attn512 N 16384 fused strides: rel-L2 {r:.3e} on {len(rows)} rows")
    assert r <= 4e-3


@pytest.mark.parametrize("shape", [(1, 1, 512, 2048), (1, 2, 256, 640)])
def test_attn512_late_dominant_key_forces_the_rescale(ldx, ldx_lib, shape):
    """One key late in the sequence dominates every row, another far earlier is its negative: the running maximum moves in the middle of a
    split's key range, O is rescaled in the accumulator file (asm on the AGPRs), and the splits' maxima differ by hundreds of log2 units in the merge."""
    td, code = DT["bf16"]
    B, H, N, M = shape
    g = torch.Generator(device="cuda").manual_seed(9)
    q = torch.randn(B, N, H * 512, device="cuda", generator=g)
    k = torch.randn(B, M, H * 512, device="cuda", generator=g)
    v = torch.randn(B, M, H * 512, device="cuda", generator=g)
    k[:, M - 70] = q[:, 17] * 4.0
    k[:, 45] = -q[:, 17] * 4.0
    q, k, v = q.to(td), k.to(td), v.to(td)
    scale = 1.0 / math.sqrt(512)
    out = _run(ldx_lib, ldx, q, k, v, H, scale, code)
    r = _rel(out, _ref(q, k, v, H, scale))
    print(f"attn512 rescale {shape}: rel-L2 {r:.3e}")
    assert r <= 4e-3


def test_vae_decode_flash_equals_the_gemm_softmax_gemm_path(ldx, ldx_lib):
    """Full-size VAE decoder (C = 512 mid block) at latent 64^2: the flash kernel against the chunked GEMM -> row softmax -> GEMM path (LDX_ATTN512=0,
    read once per process -> subprocess).  Same math, other roundings (P is rounded to 16 bit in both; the old path also rounds S): images within
    1e-2 rel-L2 (measured 5.3e-3), and the flash plan has fewer launches (no V^T GEMM, no softmax_rows)."""
    code = textwrap.dedent('''
        import sys, torch, numpy as np
        sys.path.insert(0, %r)
        import ldx_amd as ldx
        cfg = ldx.VAEConfig()
        sd = ldx.weights.synth_state_dict(ldx.weights.vae_decoder_state_dict_spec(cfg), seed=1, dtype=torch.float32)
        eng = ldx.VAEDecoderEngine(cfg, sd, dtype="bf16")
        z = torch.randn(1, 4, 64, 64, generator=torch.Generator().manual_seed(2)).cuda()
        img = eng.decode(z).float().cpu().numpy()
        np.save(sys.argv[1], img)
        print("launches", eng.plan_info()["launches"])
    ''') % ROOT
    imgs, launches = {}, {}
    import numpy as np
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        for mode in ("1", "0"):
            path = os.path.join(td, f"img{mode}.npy")
            r = subprocess.run([sys.executable, "-c", code, path], env=dict(os.environ, LDX_ATTN512=mode), capture_output=True, text=True, timeout=900)
            assert r.returncode == 0, r.stderr[-2000:]
            imgs[mode] = np.load(path)
            launches[mode] = int([l for l in r.stdout.splitlines() if l.startswith("launches")][0].split()[-1])
    rel = float(np.linalg.norm(imgs["1"] - imgs["0"]) / np.linalg.norm(imgs["0"]))
    print(f"VAE decode 512^2 flash vs GEMM/softmax/GEMM: rel-L2 {rel:.3e}; launches {launches}")
    assert rel <= 1e-2
    assert launches["1"] < launches["0"]
