"""Pin the oracle (oracle/sd15_oracle.py) against outputs of the reference itself.

tests/golden/*.npz were produced by oracle/ref_capture.py, which imports /root/reference in the build
container (the reference has no tests or golden vectors of its own).  CPU only.
Tolerances (fp32 vs fp32, different op order only): schedules exact / 1e-6; UNet forward rtol 1e-4;
20-step latents rtol 1e-3 of the latent scale.
"""
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
def sched(golden_dir):
    return np.load(os.path.join(golden_dir, "schedules.npz"))


def test_sigma_tables_exact(sched):
    assert np.array_equal(O.SIGMAS.numpy(), sched["sigmas"])
    assert np.array_equal(O.LOG_SIGMAS.numpy(), sched["log_sigmas"])
    # KATs recorded in SURVEY.md §8 a1
    assert float(O.SIGMAS[0]) == 0.029167158529162407 and float(O.SIGMAS[-1]) == 14.614641189575195


@pytest.mark.parametrize("name", ["karras", "normal", "simple", "beta"])
@pytest.mark.parametrize("steps", [1, 8, 20, 28])
def test_schedules(sched, name, steps):
    got = O.calculate_sigmas(name, steps).numpy()
    want = sched[f"{name}_{steps}"]
    assert got.shape == want.shape                 # beta dedups -> may be shorter than steps+1
    np.testing.assert_allclose(got, want, rtol=1e-6, atol=0)


@pytest.mark.parametrize("key,name,steps,den", [("karras_10_d0.45", "karras", 10, 0.45), ("normal_8_d0.3", "normal", 8, 0.3)])
def test_denoise_truncation(sched, key, name, steps, den):
    np.testing.assert_allclose(O.sigmas_for(name, steps, den).numpy(), sched[key], rtol=1e-6)


def test_timestep_lookup_exact(sched):
    got = O.timestep(torch.from_numpy(sched["timestep_in"])).numpy()
    assert np.array_equal(got, sched["timestep_out"])          # integer work: bit exact


@pytest.fixture(scope="module", params=[32, 64])
def tiny(request, golden_dir, ldx):
    mcn = request.param
    cfg = ldx.UNetConfig.tiny(mcn, {32: 64, 64: 128}[mcn])
    sd = ldx.weights.synth_state_dict(ldx.weights.unet_state_dict_spec(cfg), seed=1234)
    return cfg, sd, np.load(os.path.join(golden_dir, f"unet_mc{mcn}.npz"))


def test_unet_forward(tiny):
    cfg, sd, g = tiny
    with torch.no_grad():
        y = O.unet_forward(sd, cfg, torch.from_numpy(g["unet_x"]), torch.from_numpy(g["unet_t"]), torch.from_numpy(g["unet_ctx"]))
        yo = O.unet_forward(sd, cfg, torch.from_numpy(g["odd_x"]), torch.tensor([500.0]), torch.from_numpy(g["unet_ctx"][:1]))
    assert _rel(y, g["unet_y"]) < 1e-4
    assert _rel(yo, g["odd_y"]) < 1e-4                          # odd sizes: ceil-downsample, resize-to-skip


def test_apply_model(tiny):
    cfg, sd, g = tiny
    with torch.no_grad():
        d = O.apply_model(sd, cfg, torch.from_numpy(g["unet_x"]), torch.from_numpy(g["am_sigma"]), torch.from_numpy(g["unet_ctx"]))
    assert _rel(d, g["am_out"]) < 1e-4


def test_wrapper_contract(tiny):
    """What crosses model_options['model_function_wrapper'] (cond.py:254-263): [uncond;cond] order,
    sigma values as timestep, lcm-padded context — and the oracle reproduces `out` from those inputs."""
    cfg, sd, g = tiny
    assert int(g["hook_n"]) == 2
    for i in range(2):
        x, ts, ctx = (torch.from_numpy(g[f"hook{i}_{k}"]) for k in ("input", "timestep", "ctx"))
        assert list(g[f"hook{i}_cou"]) == [1, 0]
        assert x.shape == (6, 4, 8, 8) and ctx.shape[1] == 154 and ts.shape == (6,)
        assert torch.equal(x[:3], x[3:]) and torch.all(ts == ts[0])
        N, P = torch.from_numpy(g["N"]), torch.from_numpy(g["P"])
        assert torch.equal(ctx[:3], N.expand(3, -1, -1)) and torch.equal(ctx[3:], P.repeat(1, 2, 1).expand(3, -1, -1))
        with torch.no_grad():
            out = O.apply_model(sd, cfg, x, ts, ctx)
        assert _rel(out, g[f"hook{i}_out"]) < 1e-4


RUNS = {
    "euler_ms_off": dict(sampler_name="sample_euler", scheduler="normal", enable_multiscale=False),
    "euler_ms_on": dict(sampler_name="sample_euler", scheduler="normal", enable_multiscale=True),
    "euler_forced": dict(sampler_name="euler", scheduler="karras", enable_multiscale=False),
    "dpmpp2m": dict(sampler_name="dpmpp_2m_cfgpp", scheduler="karras", enable_multiscale=False),
}


@pytest.mark.parametrize("name", list(RUNS))
def test_ksampler_end_to_end(tiny, name):
    cfg, sd, g = tiny
    P, N = torch.from_numpy(g["P"]), torch.from_numpy(g["N"])
    trace = []
    with torch.no_grad():
        out = O.ksampler_sample(lambda x, s, c: O.apply_model(sd, cfg, x, s, c), seed=42, steps=20, cfg=7.0,
                                positive=P, negative=N, latent_image=torch.zeros(1, 4, 16, 16), trace=trace, **RUNS[name])
    assert [t[-1] for t in trace] == list(g[f"ks_{name}_res"])    # multiscale step pattern incl. whitelist quirk
    assert _rel(out, g[f"ks_{name}"]) < 1e-3


def test_ksampler_img2img(tiny):
    cfg, sd, g = tiny
    P, N = torch.from_numpy(g["P"]), torch.from_numpy(g["N"])
    with torch.no_grad():
        out = O.ksampler_sample(lambda x, s, c: O.apply_model(sd, cfg, x, s, c), seed=3, steps=10, cfg=5.0, denoise=0.45,
                                positive=P, negative=N, latent_image=torch.from_numpy(g["ks_img2img_latent"]),
                                sampler_name="sample_euler", scheduler="normal", enable_multiscale=False)
    assert _rel(out, g["ks_img2img"]) < 1e-3


def test_full_width_20_steps_config1_vs_reference(ldx, golden_dir):
    """BASELINE config 1 END TO END at full width (859.5 M parameters): the reference's own KSampler.sample latents for 20 sample_euler / normal
    steps at 512x512 (oracle/ref_capture_full20.py, ks64_20) against the oracle's sampler chain — ~50 s of CPU.  The per-step rms trace
    pins every one of the 20 model calls, not only the final latents."""
    g = np.load(os.path.join(golden_dir, "unet_full20.npz"))
    cfg = ldx.UNetConfig.sd15()
    sd = {k: v.float() for k, v in ldx.weights.synth_state_dict(ldx.weights.unet_state_dict_spec(cfg), seed=1234).items()}
    rms = []

    def model(x, s, c):
        o = O.apply_model(sd, cfg, x, s, c)
        rms.append(float(o.float().pow(2).mean().sqrt()))
        return o

    with torch.no_grad():
        out = O.ksampler_sample(model, seed=42, steps=20, cfg=7.0, positive=torch.from_numpy(g["P"]), negative=torch.from_numpy(g["N"]),
                                latent_image=torch.zeros(1, 4, 64, 64), sampler_name="sample_euler", scheduler="normal", enable_multiscale=False)
    r = _rel(out, g["ks64_20_out"])
    print(f"oracle vs reference, 20 steps at 64^2 full width: rel-L2 {r:.3e}")
    assert r < 1e-3
    assert len(rms) == 20
    np.testing.assert_allclose(np.array(rms), g["ks64_20_trace"], rtol=1e-3)


def test_state_dict_layout_matches_survey(ldx):
    spec = ldx.weights.unet_state_dict_spec(ldx.UNetConfig.sd15())
    assert len(spec) == 686 and ldx.weights.param_count(spec) == 859_520_964
