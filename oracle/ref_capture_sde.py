"""Golden capture for dpmpp_sde_cfgpp by importing the reference.  torchsde is absent offline, so the
reference's BrownianTreeNoiseSampler cannot run; it is replaced, for the capture only, by the build's stand-in
(sd15_oracle.BrownianIntervalNoise drawing from the global CPU RNG) — the goldens therefore pin the reference's SAMPLER
ARITHMETIC (two evaluations per step, ancestral splits, multi-scale flags, final Euler step) with that noise injected,
not torchsde's noise values.  Build container only; writes tests/golden/sde.npz.  See oracle/ref_capture.py."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "oracle"))
import ref_capture  # noqa: E402
import sd15_oracle as O  # noqa: E402


def main():
    sys.path.insert(0, REPO)
    import ldx_amd as ldx
    torch.set_num_threads(8)
    ref_capture.enter_reference()
    OUT = ref_capture.OUT
    from src.sample import sampling, sampling_util

    class Patched:                               # signature of sampling_util.BrownianTreeNoiseSampler (:253-255)
        def __init__(self, x, sigma_min, sigma_max, seed=None, transform=lambda x: x, cpu=False):
            self.inner = O.BrownianIntervalNoise(x, seed)

        def __call__(self, sigma, sigma_next):
            return self.inner(sigma, sigma_next)

    sampling_util.BrownianTreeNoiseSampler = Patched
    g = {}
    mcn, ctxd, lat = 64, 128, 16
    cfg = ldx.UNetConfig.tiny(mcn, ctxd)
    sd = ldx.weights.synth_state_dict(ldx.weights.unet_state_dict_spec(cfg), seed=1234)
    model, mp = ref_capture.build_reference_model(cfg, sd)
    g7 = torch.Generator().manual_seed(7)
    P = torch.randn([1, 77, ctxd], generator=g7)
    N = torch.randn([1, 77, ctxd], generator=g7)
    z = torch.zeros(1, ctxd)
    pos, neg = [[P, {"pooled_output": z}]], [[N, {"pooled_output": z}]]
    g["P"], g["N"] = P.numpy(), N.numpy()
    gl = torch.randn([2, 4, lat, lat], generator=torch.Generator().manual_seed(3)) * 0.5
    K = sampling.KSampler
    with torch.no_grad():
        # txt2img, 20 steps karras: multi-scale on by default (steps 3..11 at half resolution, both evaluations of a step)
        a = K().sample(model=mp, seed=11, steps=20, cfg=7.0, denoise=1.0, positive=pos, negative=neg,
                       latent_image={"samples": torch.zeros(1, 4, lat, lat)}, pipeline=True, disable_pbar=True,
                       sampler_name="dpmpp_sde_cfgpp", scheduler="karras")
        # multi-scale off, batch 2 img2img
        b = K().sample(model=mp, seed=12, steps=8, cfg=6.0, denoise=0.6, positive=pos, negative=neg,
                       latent_image={"samples": gl}, pipeline=True, disable_pbar=True,
                       sampler_name="dpmpp_sde_cfgpp", scheduler="normal", enable_multiscale=False)
    g["sde_txt2img"] = a[0]["samples"].numpy()
    g["sde_latent"] = gl.numpy(); g["sde_img2img"] = b[0]["samples"].numpy()
    np.savez_compressed(os.path.join(OUT, "sde.npz"), **g)
    print("sde.npz", {k: v.shape for k, v in g.items()})


if __name__ == "__main__":
    main()
