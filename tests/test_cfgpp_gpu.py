"""euler_cfgpp (+ dy extra steps) on the UNet engine and the Flux sampling chain (ModelSamplingFlux / CONST / Flux1 latent
format / FluxCFGDenoiser) on a real MI355X, vs the reference's KSampler goldens (tests/golden/cfgpp.npz).

Tolerances as in test_engine_gpu.py: sampler latents rel-L2 <= 1e-2 (fp16 engine) / 5e-2 (bf16)."""
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
    return np.load(os.path.join(golden_dir, "cfgpp.npz"))


def test_flux_schedules_match_reference(ldx, g):
    ms = ldx.sampling.ModelSamplingFlux()
    for sched in ("beta", "simple"):
        for steps in (4, 20, 28):
            assert np.allclose(ldx.sampling.calculate_sigmas(ms, sched, steps).numpy(), g[f"flux_{sched}_{steps}"], rtol=1e-6)
    with pytest.raises(AttributeError):
        ldx.sampling.calculate_sigmas(ms, "normal", 20)


@pytest.mark.parametrize("dt,tol", [("f16", 1e-2), ("bf16", 5e-2)])
def test_sd_euler_cfgpp(ldx, ldx_lib, g, dt, tol):
    cfg = ldx.UNetConfig.tiny(64, 128)
    sd = ldx.weights.synth_state_dict(ldx.weights.unet_state_dict_spec(cfg), seed=1234)
    ks = ldx.sampling.KSampler(ldx.UNetEngine(cfg, sd, device=0, dtype=dt))
    P, N = torch.from_numpy(g["P"]), torch.from_numpy(g["N"])
    trace = []
    out = ks.sample(seed=21, steps=8, cfg=7.0, positive=P, negative=N, latent_image=torch.zeros(1, 4, 16, 16), sampler_name="euler_cfgpp",
                    scheduler="karras", trace=trace)
    assert [t[-1] for t in trace] == [int(v) for v in g["sd_cfgpp_calls"][:, 1]]
    r1 = _rel(out, g["sd_cfgpp"])
    out = ks.sample(seed=22, steps=6, cfg=1.0, positive=P, negative=N, latent_image=torch.zeros(2, 4, 18, 14), sampler_name="euler_cfgpp",
                    scheduler="normal")
    r2 = _rel(out, g["sd_cfgpp_cfg1"])
    print(f"[{dt}] euler_cfgpp: cfg7 rel-L2 {r1:.3e}, cfg1 batch-2 odd-size {r2:.3e}")
    assert r1 <= tol and r2 <= tol


@pytest.mark.parametrize("dt,tol", [("f16", 1e-2), ("bf16", 5e-2)])
def test_flux_ksampler(ldx, ldx_lib, g, dt, tol):
    cfg = ldx.FluxConfig.tiny()
    sd = ldx.weights.synth_state_dict(ldx.weights.flux_state_dict_spec(cfg), seed=31, dtype=torch.float32)
    ks = ldx.sampling.FluxKSampler(ldx.FluxEngine(cfg, sd, device=0, dtype=dt))
    ctx, y = torch.from_numpy(g["flux_ctx"]), torch.from_numpy(g["flux_y"])
    neg = (torch.zeros_like(ctx), torch.zeros_like(y))                     # ConditioningZeroOut (pipeline.py:247-249)
    trace = []
    out = ks.sample(seed=9, steps=6, cfg=1, sampler_name="euler_cfgpp", scheduler="beta", positive=(ctx, y), negative=neg,
                    latent_image=torch.zeros(1, 16, 8, 12), guidance=3.0, trace=trace)
    assert [t[-1] for t in trace] == [int(v) for v in g["flux_ks_calls"][:, 1]]
    r1 = _rel(out, g["flux_ks"])
    out = ks.sample(seed=10, steps=5, cfg=1, sampler_name="sample_euler", scheduler="simple", positive=(ctx, y), negative=neg,
                    latent_image=torch.from_numpy(g["flux_i2i_latent"]), guidance=3.0, denoise=0.6)
    r2 = _rel(out, g["flux_i2i"])
    print(f"[{dt}] Flux KSampler: euler_cfgpp/beta rel-L2 {r1:.3e}, img2img euler/simple {r2:.3e}")
    assert r1 <= tol and r2 <= tol


@pytest.mark.parametrize("dt,tol", [("f16", 1e-2), ("bf16", 5e-2)])
def test_flux_first_block_cache(ldx, ldx_lib, g, dt, tol):
    """Opt-in approximate mode (ldx_flux_fbcache): same number of cache hits as the reference's patched model and latents
    within the sampler tolerance of ITS (approximate) output.  Threshold 0.9 mixes hits and misses on the synthetic weights;
    0.12 (the pipeline's value) never hits here and must equal the exact path."""
    cfg = ldx.FluxConfig.tiny()
    sd = ldx.weights.synth_state_dict(ldx.weights.flux_state_dict_spec(cfg), seed=31, dtype=torch.float32)
    eng = ldx.FluxEngine(cfg, sd, device=0, dtype=dt)
    ks = ldx.sampling.FluxKSampler(eng)
    ctx, y = torch.from_numpy(g["flux_ctx"]), torch.from_numpy(g["flux_y"])
    neg = (torch.zeros_like(ctx), torch.zeros_like(y))
    for thr, tag in ((0.9, "t90"), (0.12, "t12")):
        eng.set_fbcache(thr)
        out = ks.sample(seed=9, steps=12, cfg=1, sampler_name="euler_cfgpp", scheduler="beta", positive=(ctx, y), negative=neg,
                        latent_image=torch.zeros(1, 16, 8, 12), guidance=3.0)
        st = eng.fbcache_stats()
        ref_hits = int(g[f"fb_{tag}_hits"].sum())
        r1 = _rel(out, g[f"fb_{tag}_out"])
        eng.set_fbcache(thr)
        out = ks.sample(seed=10, steps=10, cfg=1, sampler_name="sample_euler", scheduler="simple", positive=(ctx, y), negative=neg,
                        latent_image=torch.zeros(2, 16, 8, 8), guidance=3.0)
        st2 = eng.fbcache_stats()
        r2 = _rel(out, g[f"fb_{tag}_euler_out"])
        print(f"[{dt}] FBCache thr {thr}: hits {st['hits']}/{st['hits'] + st['misses']} (ref {ref_hits}), rel-L2 {r1:.3e}; "
              f"euler hits {st2['hits']} (ref {int(g[f'fb_{tag}_euler_hits'].sum())}), rel-L2 {r2:.3e}")
        assert st["hits"] == ref_hits and st["hits"] + st["misses"] == len(g[f"fb_{tag}_hits"])
        assert st2["hits"] == int(g[f"fb_{tag}_euler_hits"].sum())
        assert r1 <= tol and r2 <= tol
    eng.set_fbcache(0.0)
