"""Pin the euler_cfgpp (+ dy extra step) and Flux-sampling restatements against the reference's KSampler.

tests/golden/cfgpp.npz was produced by oracle/ref_capture_flux_sampler.py (imports /root/reference in the build
container).  CPU only.  Tolerances: tables 1e-6 relative (float32 pow/exp order), latents rel-L2 1e-3."""
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
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "cfgpp.npz"))


def test_flux_sigma_table_and_schedules(g):
    t = O.flux_model_sigmas()
    assert t.shape == (10000,) and t.dtype == torch.float32
    assert np.allclose(t[:8].numpy(), g["flux_sigmas_head"], rtol=1e-6) and np.allclose(t[-8:].numpy(), g["flux_sigmas_tail"], rtol=1e-6)
    assert np.allclose(t[::499].numpy(), g["flux_sigmas_sample"], rtol=1e-6)
    for sched in ("beta", "simple"):
        for steps in (4, 20, 28):
            assert np.allclose(O.calculate_sigmas(sched, steps, t).numpy(), g[f"flux_{sched}_{steps}"], rtol=1e-6), (sched, steps)
    with pytest.raises(AttributeError):           # the reference's ModelSamplingFlux has no sigma_min
        O.calculate_sigmas("normal", 20, t)


@pytest.fixture(scope="module")
def tiny(ldx):
    cfg = ldx.UNetConfig.tiny(64, 128)
    sd = ldx.weights.synth_state_dict(ldx.weights.unet_state_dict_spec(cfg), seed=1234)
    return cfg, sd


def test_sd_euler_cfgpp(g, tiny):
    cfg, sd = tiny
    P, N = torch.from_numpy(g["P"]), torch.from_numpy(g["N"])
    den = lambda x, s, c: O.apply_model(sd, cfg, x, s, c)      # noqa: E731
    trace = []
    with torch.no_grad():
        out = O.ksampler_sample(den, seed=21, steps=8, cfg=7.0, positive=P, negative=N, latent_image=torch.zeros(1, 4, 16, 16),
                                sampler_name="euler_cfgpp", scheduler="karras", trace=trace)
    assert [t[-1] for t in trace] == [int(v) for v in g["sd_cfgpp_calls"][:, 1]]        # dy half-res calls after steps 2 and 3
    assert _rel(out, g["sd_cfgpp"]) < 1e-3
    trace = []
    with torch.no_grad():                                      # cfg 1: both branches still evaluated; odd/even mix 18x14, batch 2
        out = O.ksampler_sample(den, seed=22, steps=6, cfg=1.0, positive=P, negative=N, latent_image=torch.zeros(2, 4, 18, 14),
                                sampler_name="euler_cfgpp", scheduler="normal", trace=trace)
    assert [t[-1] for t in trace] == [int(v) for v in g["sd_cfgpp_cfg1_calls"][:, 1]]
    assert _rel(out, g["sd_cfgpp_cfg1"]) < 1e-3


@pytest.fixture(scope="module")
def flux(ldx):
    cfg = ldx.FluxConfig.tiny()
    sd = ldx.weights.synth_state_dict(ldx.weights.flux_state_dict_spec(cfg), seed=31, dtype=torch.float32)
    return cfg, sd


def test_flux_ksampler(g, flux):
    cfg, sd = flux
    ctx, y = torch.from_numpy(g["flux_ctx"]), torch.from_numpy(g["flux_y"])
    den = lambda x, s, c, yy, gd: O.flux_apply_model(sd, cfg, x, s, c, yy, gd)      # noqa: E731
    trace = []
    with torch.no_grad():
        out = O.flux_ksampler_sample(den, seed=9, steps=6, cfg=1, sampler_name="euler_cfgpp", scheduler="beta", positive=(ctx, y),
                                     negative=(torch.zeros_like(ctx), torch.zeros_like(y)), latent_image=torch.zeros(1, 16, 8, 12),
                                     guidance=3.0, trace=trace)
    assert [t[-1] for t in trace] == [int(v) for v in g["flux_ks_calls"][:, 1]]
    assert _rel(out, g["flux_ks"]) < 1e-3
    with torch.no_grad():
        out = O.flux_ksampler_sample(den, seed=10, steps=5, cfg=1, sampler_name="sample_euler", scheduler="simple", positive=(ctx, y),
                                     negative=(torch.zeros_like(ctx), torch.zeros_like(y)), latent_image=torch.from_numpy(g["flux_i2i_latent"]),
                                     guidance=3.0, denoise=0.6)
    assert _rel(out, g["flux_i2i"]) < 1e-3


@pytest.mark.parametrize("thr,tag", [(0.9, "t90"), (0.12, "t12")])
def test_flux_first_block_cache(g, flux, thr, tag):
    """WaveSpeed FBCache restatement: identical hit / miss decisions and latents as the reference's patched model."""
    cfg, sd = flux
    ctx, y = torch.from_numpy(g["flux_ctx"]), torch.from_numpy(g["flux_y"])
    neg = (torch.zeros_like(ctx), torch.zeros_like(y))
    fb = O.FluxFBCache(thr)
    den = lambda x, s, c, yy, gd: O.flux_apply_model(sd, cfg, x, s, c, yy, gd, fb=fb)      # noqa: E731
    with torch.no_grad():
        out = O.flux_ksampler_sample(den, seed=9, steps=12, cfg=1, sampler_name="euler_cfgpp", scheduler="beta", positive=(ctx, y),
                                     negative=neg, latent_image=torch.zeros(1, 16, 8, 12), guidance=3.0)
    assert fb.log == [int(v) for v in g[f"fb_{tag}_hits"]]
    assert _rel(out, g[f"fb_{tag}_out"]) < 1e-3
    fb.log.clear(); fb.reset()
    with torch.no_grad():
        out = O.flux_ksampler_sample(den, seed=10, steps=10, cfg=1, sampler_name="sample_euler", scheduler="simple", positive=(ctx, y),
                                     negative=neg, latent_image=torch.zeros(2, 16, 8, 8), guidance=3.0)
    assert fb.log == [int(v) for v in g[f"fb_{tag}_euler_hits"]]
    assert _rel(out, g[f"fb_{tag}_euler_out"]) < 1e-3
