"""Pin the Flux DiT oracle restatement against the reference's own Flux3 module (tiny config, CPU only).
Goldens: oracle/ref_capture_flux.py."""
import os

import numpy as np
import pytest
import torch

from oracle import sd15_oracle as O


def _rel(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.fixture(scope="module")
def flux(ldx, golden_dir):
    cfg = ldx.FluxConfig.tiny()
    sd = ldx.weights.synth_state_dict(ldx.weights.flux_state_dict_spec(cfg), seed=31, dtype=torch.float32)
    return cfg, sd, np.load(os.path.join(golden_dir, "flux.npz"))


@pytest.mark.parametrize("case", ["a", "b"])
def test_flux_forward_oracle(flux, case):
    cfg, sd, g = flux
    T = lambda k: torch.from_numpy(g[f"{case}_{k}"])
    with torch.no_grad():
        out = O.flux_forward(sd, cfg, T("x"), T("t"), T("ctx"), T("y"), T("g"))
    assert _rel(out, g[f"{case}_out"]) < 1e-4


def test_rope_tables_match_reference(ldx, flux):
    """Host-built cos/sin tables (engine.flux_rope_tables) against the reference's pe_embedder output."""
    cfg, sd, g = flux
    pe = torch.from_numpy(g["pe_probe"])[0, 0]                 # [3 tokens][16 pairs][2][2] for ids (0,0,0),(0,1,2),(0,2,1)
    cos, sin = ldx.engine.flux_rope_tables(cfg, 0, 6, 6)       # img tokens of a 3x3 grid: token (r, c) = r*3 + c
    for tok, (r, c) in enumerate([(0, 0), (1, 2), (2, 1)]):
        i = r * 3 + c
        assert torch.equal(pe[tok, :, 0, 0], cos[i]) and torch.equal(pe[tok, :, 1, 0], sin[i])
        assert torch.equal(pe[tok, :, 0, 1], -sin[i]) and torch.equal(pe[tok, :, 1, 1], cos[i])


def test_flux_dev_layout(ldx):
    spec = ldx.weights.flux_state_dict_spec(ldx.FluxConfig())
    n = ldx.weights.param_count(spec)
    assert abs(n - 11.90e9) / 11.90e9 < 0.01                    # 11.90 B parameters
