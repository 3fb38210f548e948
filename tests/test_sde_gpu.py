"""dpmpp_sde_cfgpp through the HIP engine (ldx_unet_denoise + ldx_sampler_step + ldx_bilinear) on a real MI355X vs the goldens
captured from the reference's sampler with the Brownian stand-in injected (oracle/ref_capture_sde.py).
Tolerances as for the other 20-step sampler goldens: fp16-activation mode rel-L2 <= 1e-2; bf16 <= 7e-2 (39 model
evaluations at cfg 7 with injected noise amplify the per-forward bf16 spread like the ancestral sampler does)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.double().cpu(), torch.as_tensor(b).double()
    return float((a - b).norm() / b.norm())


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "sde.npz"))


@pytest.fixture(scope="module")
def unet(ldx, ldx_lib):
    cfg = ldx.UNetConfig.tiny(64, 128)
    sd = ldx.weights.synth_state_dict(ldx.weights.unet_state_dict_spec(cfg), seed=1234)
    return {dt: ldx.UNetEngine(cfg, sd, device=0, dtype=dt) for dt in ("f16", "bf16")}


@pytest.mark.parametrize("dt,tol", [("f16", 1e-2), ("bf16", 7e-2)])
def test_dpmpp_sde_cfgpp(ldx, g, unet, dt, tol):
    ks = ldx.sampling.KSampler(unet[dt])
    P, N = torch.from_numpy(g["P"]), torch.from_numpy(g["N"])
    trace = []
    out = ks.sample(seed=11, steps=20, cfg=7.0, denoise=1.0, positive=P, negative=N, latent_image=torch.zeros(1, 4, 16, 16),
                    sampler_name="dpmpp_sde_cfgpp", scheduler="karras", trace=trace)
    r1 = _rel(out, g["sde_txt2img"])
    assert trace[:6] == [(16, 16)] * 6 and trace[6:24] == [(8, 8)] * 18 and trace[24:] == [(16, 16)] * 15
    out = ks.sample(seed=12, steps=8, cfg=6.0, denoise=0.6, positive=P, negative=N, latent_image=torch.from_numpy(g["sde_latent"]),
                    sampler_name="dpmpp_sde_cfgpp", scheduler="normal", enable_multiscale=False)
    r2 = _rel(out, g["sde_img2img"])
    print(f"[{dt}] dpmpp_sde_cfgpp txt2img {r1:.3e} img2img {r2:.3e}")
    assert r1 <= tol and r2 <= tol


def test_graph_is_replayed_across_sde_and_multiscale_steps(ldx, g, unet):
    """Round-3 advisor finding: the engine's captured graph is tied to the pointer of x, and dpmpp_sde_cfgpp alternates x / x2 while the multi-scale
    steps hand over fresh _bilinear tensors, so graph mode silently fell back to eager launches or re-captured on most evaluations.  CFGDenoiser now
    stages such inputs through one persistent tensor per shape: over a 20-step run (39 evaluations, two resolutions) the graph must be captured a
    handful of times (once per shape, plus the re-capture when a shape's input moves to the staging tensor) and replayed for nearly everything
    else — and the latents must equal the eager run's bit for bit."""
    e = unet["f16"]
    ks = ldx.sampling.KSampler(e)
    P, N = torch.from_numpy(g["P"]), torch.from_numpy(g["N"])
    kw = dict(seed=11, steps=20, cfg=7.0, denoise=1.0, positive=P, negative=N, latent_image=torch.zeros(1, 4, 16, 16),
              sampler_name="dpmpp_sde_cfgpp", scheduler="karras")
    eager = ks.sample(**kw).clone()
    c0, r0 = e.graph_stats()
    e.set_graph_mode(True)
    try:
        trace = []
        out = ks.sample(trace=trace, **kw)
    finally:
        e.set_graph_mode(False)
    c1, r1 = e.graph_stats()
    captures, replays, evals = c1 - c0, r1 - r0, len(trace)
    print(f"dpmpp_sde_cfgpp graph mode: {evals} evaluations, {captures} captures, {replays} replays")
    assert torch.equal(out, eager)
    assert evals == 39 and captures <= 4 and replays >= evals - 10, (evals, captures, replays)
    # round 4: the device buffers (context, batch, staging copy of x) live with the engine per shape, so a SECOND sampling run — new CFGDenoiser, new
    # latent and context tensors — presents the same pointers: nothing is captured again, every evaluation replays
    e.set_graph_mode(True)
    try:
        trace2 = []
        out2 = ks.sample(trace=trace2, **kw)
    finally:
        e.set_graph_mode(False)
    c2, r2 = e.graph_stats()
    print(f"second run: {len(trace2)} evaluations, {c2 - c1} captures, {r2 - r1} replays")
    assert torch.equal(out2, eager)
    assert c2 - c1 == 0 and r2 - r1 == len(trace2), (c2 - c1, r2 - r1, len(trace2))


def test_two_denoisers_on_one_engine_do_not_share_a_context(ldx, g, unet):
    """The context / batch buffers of CFGDenoiser live with the engine per shape (round 4).  Two denoisers of the same shape with DIFFERENT prompts,
    used alternately, must each see their own context: the shared buffer is restored from the private copy when its owner changed."""
    e = unet["f16"]
    P, N = torch.from_numpy(g["P"]), torch.from_numpy(g["N"])
    d1 = ldx.sampling.CFGDenoiser(e, P, N, 7.0, 1, 16, 16)
    d2 = ldx.sampling.CFGDenoiser(e, P * 0.5 + 0.1, N, 7.0, 1, 16, 16)
    x = torch.randn(1, 4, 16, 16, generator=torch.Generator().manual_seed(3)).cuda()
    a1 = [t.clone() for t in d1(x, 3.0)]
    a2 = [t.clone() for t in d2(x, 3.0)]
    b1 = [t.clone() for t in d1(x, 3.0)]
    b2 = [t.clone() for t in d2(x, 3.0)]
    assert torch.equal(a1[1], b1[1]) and torch.equal(a2[1], b2[1])
    assert not torch.equal(a1[1], a2[1])
