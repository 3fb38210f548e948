"""T5 text encoder through libldx.so (ldx_t5_encode) on a real MI355X vs the reference goldens and the oracle, and the
biased-attention op on its own.

Tolerances (16-bit activations through 3 blocks): rel-L2 <= 4e-3 (fp16) / 2.5e-2 (bf16), as for CLIP."""
import ctypes as C
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import sd15_oracle as O  # noqa: E402  (checker only)


def _rel(a, b):
    a, b = a.double().cpu(), torch.as_tensor(b).double()
    return float((a - b).norm() / b.norm())


@pytest.fixture(scope="module")
def setup(ldx, ldx_lib, golden_dir):
    cfg = ldx.T5Config.tiny()
    sd = ldx.weights.synth_state_dict(ldx.weights.t5_state_dict_spec(cfg), seed=555)
    return cfg, sd, np.load(os.path.join(golden_dir, "t5.npz"))


@pytest.mark.parametrize("dt,tol", [("f16", 4e-3), ("bf16", 2.5e-2)])
def test_t5_vs_reference_golden(ldx, setup, dt, tol):
    cfg, sd, g = setup
    eng = ldx.T5Engine(cfg, sd, device=0, dtype=dt)
    for name in ("a", "b", "c", "d"):
        out = eng.forward(torch.from_numpy(g[f"ids_{name}"]))
        r = _rel(out, g[f"out_{name}"])
        print(f"[{dt}] T5 case {name} {g[f'ids_{name}'].shape}: rel-L2 {r:.3e}")
        assert out.shape == g[f"out_{name}"].shape and r <= tol
    pairs = [list(zip([int(t) for t in g["tw_ids"]], [float(w) for w in g["tw_wts"]]))]
    cond, pooled = eng.encode_token_weights(pairs)
    assert pooled is None and _rel(cond, g["tw_cond"]) <= tol


def test_t5_wider_config_vs_oracle(ldx, ldx_lib):
    """Head dim 64 as in T5-XXL, d_ff not a multiple of 128, ragged L."""
    cfg = ldx.T5Config(d_model=512, d_ff=1344, num_layers=2, num_heads=8, vocab_size=1000)
    sd = ldx.weights.synth_state_dict(ldx.weights.t5_state_dict_spec(cfg), seed=9)
    eng = ldx.T5Engine(cfg, sd, device=0, dtype="f16")
    ids = torch.randint(0, 1000, (2, 77), generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        ref = O.t5_forward(sd, cfg, ids)
    assert _rel(eng.forward(ids), ref) <= 4e-3


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("B,H,N,D", [(2, 4, 256, 64), (1, 8, 77, 40), (1, 2, 300, 32)])
def test_attention_with_bias(ldx, ldx_lib, dt, B, H, N, D):
    td = torch.bfloat16 if dt == "bf16" else torch.float16
    code = 0 if dt == "bf16" else 1
    L = ldx_lib
    g = torch.Generator(device="cuda").manual_seed(N + D)
    Cc = H * D
    qkv = torch.randn(B, N, 3 * Cc, device="cuda", generator=g).to(td)
    lp = (N + 63) // 64 * 64
    bias = torch.zeros(H, N, lp, device="cuda")
    bias[:, :, :N] = torch.randn(H, N, N, device="cuda", generator=g) * 2.0
    out = torch.empty(B, N, Cc, device="cuda", dtype=td)
    p = lambda t: C.c_void_p(t.data_ptr())      # noqa: E731
    scale = 1.0 / math.sqrt(D)
    # the ABI adds the bias before the scale: pass bias / scale to get  q.k * scale + bias
    bs = (bias / scale).contiguous()
    ldx.lib.check(L.ldx_op_attention_bias(p(qkv), 3 * Cc, p(qkv[..., Cc:]), 3 * Cc, p(qkv[..., 2 * Cc:]), 3 * Cc, p(out), Cc, B, H, N, N, D,
                                          scale, p(bs), lp, N * lp, code, C.c_void_p(torch.cuda.current_stream().cuda_stream)), "attn_bias")
    q = qkv[..., :Cc].float().view(B, N, H, D).transpose(1, 2)
    k = qkv[..., Cc:2 * Cc].float().view(B, N, H, D).transpose(1, 2)
    v = qkv[..., 2 * Cc:].float().view(B, N, H, D).transpose(1, 2)
    ref = torch.softmax(q @ k.transpose(-1, -2) * scale + bias[None, :, :, :N], -1) @ v
    got = out.float().view(B, N, H, D).transpose(1, 2)
    rel = float((got - ref).norm() / ref.norm())
    assert rel <= (1e-2 if dt == "bf16" else 2e-3), rel
