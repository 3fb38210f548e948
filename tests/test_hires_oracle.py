"""Pin the HiresFix-row restatements (bislerp, VAE encode, euler_ancestral_cfgpp) against outputs of the reference.

tests/golden/hires.npz was produced by oracle/ref_capture_hires.py (imports /root/reference in the build container).
CPU only.  Tolerances: bislerp abs 2e-5 of unit-scale latents (acos/sin order only); VAE moments rel-L2 1e-4;
sampler latents rel-L2 1e-3 (same as the other KSampler goldens).
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
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "hires.npz"))


def test_bislerp(g):
    for i in range(int(g["bs_n"])):
        wn, hn = (int(v) for v in g[f"bs_wh_{i}"])
        out = O.bislerp(torch.from_numpy(g[f"bs_in_{i}"]), wn, hn)
        assert out.shape == g[f"bs_out_{i}"].shape
        assert float((out - torch.from_numpy(g[f"bs_out_{i}"])).abs().max()) <= 2e-5, i


def test_latent_upscale_floor_semantics(g):
    # LatentUpscale.upscale (upscale.py:149-166): max(64, .) // 8
    out = O.bislerp(torch.from_numpy(g["bs_in_1"]), max(64, 256) // 8, max(64, 192) // 8)
    assert float((out - torch.from_numpy(g["lu_out"])).abs().max()) <= 2e-5


def test_vae_encode(g, ldx):
    for tag in g["enc_tags"]:
        ch = int(str(tag).split("_")[0])
        cfg = ldx.VAEConfig(ch=ch)
        sd = ldx.weights.synth_state_dict(ldx.weights.vae_state_dict_spec(cfg), seed=4321, dtype=torch.float32)
        with torch.no_grad():
            mom = O.vae_encode_moments(sd, cfg, torch.from_numpy(g[f"enc_px_{tag}"]))
            assert _rel(mom, g[f"enc_mom_{tag}"]) < 1e-4, tag
            torch.manual_seed(5)
            smp = O.vae_sample_moments(mom)
        assert _rel(smp, g[f"enc_sample_{tag}"]) < 1e-4, tag


def test_decoder_weights_unchanged_by_encoder_keys(ldx):
    """synth_state_dict seeds per key: the decoder goldens (vae.npz) stay valid for the full VAE spec."""
    cfg = ldx.VAEConfig(ch=64)
    a = ldx.weights.synth_state_dict(ldx.weights.vae_decoder_state_dict_spec(cfg), seed=4321, dtype=torch.float32)
    b = ldx.weights.synth_state_dict(ldx.weights.vae_state_dict_spec(cfg), seed=4321, dtype=torch.float32)
    assert all(torch.equal(a[k], b[k]) for k in a) and len(b) > len(a)


@pytest.fixture(scope="module")
def tiny(ldx):
    cfg = ldx.UNetConfig.tiny(64, 128)
    sd = ldx.weights.synth_state_dict(ldx.weights.unet_state_dict_spec(cfg), seed=1234)
    return cfg, sd


def test_euler_ancestral_cfgpp(g, tiny):
    cfg, sd = tiny
    P, N = torch.from_numpy(g["P"]), torch.from_numpy(g["N"])
    den = lambda x, s, c: O.apply_model(sd, cfg, x, s, c)      # noqa: E731
    with torch.no_grad():
        out = O.ksampler_sample(den, seed=5, steps=10, cfg=8.0, denoise=0.45, positive=P, negative=N,
                                latent_image=torch.from_numpy(g["anc_latent"]), sampler_name="euler_ancestral_cfgpp", scheduler="normal")
        assert _rel(out, g["anc_img2img"]) < 1e-3
        out = O.ksampler_sample(den, seed=6, steps=8, cfg=7.0, denoise=1.0, positive=P, negative=N,
                                latent_image=torch.zeros(1, 4, 16, 16), sampler_name="euler_ancestral_cfgpp", scheduler="karras")
        assert _rel(out, g["anc_txt2img"]) < 1e-3


def test_hiresfix_chain(g, tiny):
    """pipeline.py:346-366: LatentUpscale x2 of the txt2img latents, then 10 steps euler_ancestral_cfgpp / normal,
    cfg 8, denoise 0.45."""
    cfg, sd = tiny
    P, N = torch.from_numpy(g["P"]), torch.from_numpy(g["N"])
    up = O.bislerp(torch.from_numpy(g["hf_base"]), 32, 32)
    ref_up = torch.from_numpy(g["hf_up"])
    assert float((up - ref_up).abs().max()) <= 2e-6 * float(ref_up.abs().max())
    with torch.no_grad():
        out = O.ksampler_sample(lambda x, s, c: O.apply_model(sd, cfg, x, s, c), seed=77, steps=10, cfg=8, denoise=0.45,
                                positive=P, negative=N, latent_image=up, sampler_name="euler_ancestral_cfgpp", scheduler="normal")
    assert _rel(out, g["hf_out"]) < 1e-3
