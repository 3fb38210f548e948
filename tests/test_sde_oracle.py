"""dpmpp_sde_cfgpp: the oracle restatement against the goldens captured from the reference's sampler with
the build's Brownian stand-in injected (oracle/ref_capture_sde.py — torchsde is absent offline, so the noise VALUES are
the stand-in's on both sides; what is pinned is the sampler arithmetic).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import sd15_oracle as O


def _rel(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).norm() / b.norm())


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "sde.npz"))


@pytest.fixture(scope="module")
def tiny(ldx):
    cfg = ldx.UNetConfig.tiny(64, 128)
    sd = ldx.weights.synth_state_dict(ldx.weights.unet_state_dict_spec(cfg), seed=1234)
    return cfg, sd


def test_dpmpp_sde_cfgpp_vs_reference(g, tiny):
    cfg, sd = tiny
    P, N = torch.from_numpy(g["P"]), torch.from_numpy(g["N"])
    den = lambda x, s, c: O.apply_model(sd, cfg, x, s, c)      # noqa: E731
    trace = []
    with torch.no_grad():
        out = O.ksampler_sample(den, seed=11, steps=20, cfg=7.0, denoise=1.0, positive=P, negative=N,
                                latent_image=torch.zeros(1, 4, 16, 16), sampler_name="dpmpp_sde_cfgpp", scheduler="karras", trace=trace)
        assert _rel(out, g["sde_txt2img"]) < 1e-3
        # two evaluations per step except the last; steps 3..11 (fullres_start 3, fullres_end 8) at half resolution, both of them
        assert len(trace) == 2 * 19 + 1
        assert trace[:6] == [(16, 16)] * 6 and trace[6:24] == [(8, 8)] * 18 and trace[24:] == [(16, 16)] * 15
        out = O.ksampler_sample(den, seed=12, steps=8, cfg=6.0, denoise=0.6, positive=P, negative=N,
                                latent_image=torch.from_numpy(g["sde_latent"]), sampler_name="dpmpp_sde_cfgpp", scheduler="normal",
                                enable_multiscale=False)
        assert _rel(out, g["sde_img2img"]) < 1e-3


def test_brownian_stand_in_is_a_brownian_increment():
    """unit variance for any interval, and a nested query reuses the inner increment (correlation sqrt(|s-t| / |n-t|))."""
    x = torch.zeros(1, 4, 64, 64)
    ns = O.BrownianIntervalNoise(x, seed=3)
    a = ns(torch.tensor(4.0), torch.tensor(3.0))
    b = ns(torch.tensor(4.0), torch.tensor(2.0))
    c = ns(torch.tensor(2.0), torch.tensor(1.5))
    for t in (a, b, c):
        assert abs(float(t.std()) - 1.0) < 0.03 and abs(float(t.mean())) < 0.03
    corr = float((a * b).mean())
    assert abs(corr - (1.0 / 2.0) ** 0.5) < 0.04
    assert abs(float((b * c).mean())) < 0.04
    # the product class draws the same numbers
    import ldx_amd as ldx
    ns2 = ldx.sampling.BrownianIntervalNoise(x, seed=3)
    assert torch.equal(ns2(torch.tensor(4.0), torch.tensor(3.0)), a) and torch.equal(ns2(torch.tensor(4.0), torch.tensor(2.0)), b)
