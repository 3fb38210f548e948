"""Pin the T5 restatement (oracle.t5_forward) and the host-side bias-table builder against the reference's T5 class.

tests/golden/t5.npz was produced by oracle/ref_capture_t5.py (imports /root/reference in the build container).  CPU only.
Tolerances: fp32 vs fp32 (op order only): rel-L2 1e-5; bucket tables exact."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import sd15_oracle as O  # noqa: E402


def _rel(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.fixture(scope="module")
def setup(ldx, golden_dir):
    cfg = ldx.T5Config.tiny()
    sd = ldx.weights.synth_state_dict(ldx.weights.t5_state_dict_spec(cfg), seed=555)
    return cfg, sd, np.load(os.path.join(golden_dir, "t5.npz"))


@pytest.mark.parametrize("name", ["a", "b", "c", "d"])
def test_t5_forward(setup, name):
    cfg, sd, g = setup
    with torch.no_grad():
        out = O.t5_forward(sd, cfg, torch.from_numpy(g[f"ids_{name}"]))
    assert _rel(out, g[f"out_{name}"]) < 1e-5


def test_relative_position_buckets_exact(setup, ldx):
    cfg, sd, g = setup
    for l in (8, 40, 256, 300):
        ctx = torch.arange(l)[:, None]; mem = torch.arange(l)[None, :]
        assert np.array_equal(O.t5_relative_position_bucket(mem - ctx).numpy(), g[f"bucket_{l}"])
        assert np.array_equal(ldx.engine.t5_relative_position_bucket(mem - ctx).numpy(), g[f"bucket_{l}"])


def test_bias_table_layout(setup, ldx):
    """Host builder used by T5Engine: [H][L][Lp] fp32, Lp = L rounded up to 64, zero padded (include/ldx.h)."""
    cfg, sd, g = setup
    rel = sd["encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"]
    t = ldx.engine.t5_bias_table(rel, 40)
    assert t.shape == (cfg.num_heads, 40, 64) and t.dtype == torch.float32
    assert np.allclose(t[:, :, :40].numpy(), g["bias_40"][0], atol=1e-7) and float(t[:, :, 40:].abs().max()) == 0.0


def test_t5_token_weights(setup):
    cfg, sd, g = setup
    pairs = [list(zip([int(t) for t in g["tw_ids"]], [float(w) for w in g["tw_wts"]]))]
    with torch.no_grad():
        cond = O.t5_encode_token_weights(sd, cfg, pairs)
    assert _rel(cond, g["tw_cond"]) < 1e-5


def test_t5_xxl_layout(ldx):
    spec = ldx.weights.t5_state_dict_spec(ldx.T5Config())
    assert ldx.weights.param_count(spec) == 4_762_310_656           # T5-XXL encoder incl. shared embedding
