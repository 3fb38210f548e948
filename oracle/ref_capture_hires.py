"""Golden capture for the HiresFix rows by importing the reference: bislerp latent
upscale (G12), VAE encode (G13), euler_ancestral_cfgpp through KSampler (G14) and the LatentUpscale -> KSampler chain
of pipeline.py:346-366 (G15).  Build container only; writes tests/golden/hires.npz.  See oracle/ref_capture.py."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "oracle"))
import ref_capture  # noqa: E402


def main():
    sys.path.insert(0, REPO)
    import ldx_amd as ldx
    torch.set_num_threads(8)
    ref_capture.enter_reference()
    OUT = ref_capture.OUT
    from src.Utilities import upscale
    from src.AutoEncoders import VariationalAE
    from src.sample import sampling
    g = {}
    gen = torch.Generator().manual_seed(99)

    # ---- G12: bislerp --------------------------------------------------------------------------------
    cases = [((1, 4, 8, 12), 24, 16), ((2, 4, 16, 16), 32, 32), ((1, 4, 9, 7), 11, 13), ((1, 16, 6, 6), 12, 9),
             ((1, 4, 12, 12), 8, 6)]
    for i, (shape, wn, hn) in enumerate(cases):
        x = torch.randn(shape, generator=gen)
        if i == 0:
            x[0, :, 2, 3] = 0.0                      # zero-norm pixel
            x[0, :, 4, 5] = x[0, :, 4, 4]            # identical neighbours (dot > 1 - 1e-5 branch)
            x[0, :, 6, 7] = -x[0, :, 6, 6]           # polar opposites (lerp branch)
            x[0, :, 5, 2] = x[0, :, 4, 2] * 3.0      # parallel, different norms (vertical pass)
        y = upscale.bislerp(x, wn, hn)
        g[f"bs_in_{i}"] = x.numpy(); g[f"bs_out_{i}"] = y.numpy(); g[f"bs_wh_{i}"] = np.array([wn, hn])
    g["bs_n"] = np.array(len(cases))
    lu = upscale.LatentUpscale().upscale({"samples": torch.from_numpy(g["bs_in_1"])}, width=256, height=192)[0]["samples"]
    g["lu_out"] = lu.numpy()                          # max(64, .) // 8 semantics (upscale.py:149-166)

    # ---- G13: VAE encode -----------------------------------------------------------------------------
    for ch, (hh, ww) in ((64, (32, 48)), (128, (64, 32)), (64, (37, 52))):
        cfg = ldx.VAEConfig(ch=ch)
        dd = {"double_z": True, "z_channels": 4, "resolution": 256, "in_channels": 3, "out_ch": 3, "ch": ch,
              "ch_mult": [1, 2, 4, 4], "num_res_blocks": 2, "attn_resolutions": [], "dropout": 0.0}
        eng = VariationalAE.AutoencodingEngine(VariationalAE.Encoder(**dd), VariationalAE.Decoder(**dd),
                                               VariationalAE.DiagonalGaussianRegularizer())
        sd = ldx.weights.synth_state_dict(ldx.weights.vae_state_dict_spec(cfg), seed=4321, dtype=torch.float32)
        eng.load_state_dict(sd, strict=True)
        eng = eng.eval().float()
        px = torch.rand([1, hh, ww, 3], generator=gen)
        tag = f"{ch}_{hh}x{ww}"
        with torch.no_grad():
            xin = (px.movedim(-1, 1) * 2.0 - 1.0)     # VAE.encode: movedim + process_input (:735,752)
            mom, _ = eng.encode(xin, unregularized=True)
            torch.manual_seed(5)
            smp = eng.encode(xin)
        g[f"enc_px_{tag}"] = px.numpy(); g[f"enc_mom_{tag}"] = mom.numpy(); g[f"enc_sample_{tag}"] = smp.numpy()
    g["enc_tags"] = np.array(["64_32x48", "128_64x32", "64_37x52"])

    # ---- G14/G15: euler_ancestral_cfgpp + the HiresFix chain on the tiny UNet ---------------------------
    mcn, ctxd, lat = 64, 128, 16          # head dims 8/16/32: the smallest width the HIP engine accepts
    cfg = ldx.UNetConfig.tiny(mcn, ctxd)
    sd = ldx.weights.synth_state_dict(ldx.weights.unet_state_dict_spec(cfg), seed=1234)
    model, mp = ref_capture.build_reference_model(cfg, sd)
    g7 = torch.Generator().manual_seed(7)
    P = torch.randn([1, 77, ctxd], generator=g7)
    N = torch.randn([1, 77, ctxd], generator=g7)
    z = torch.zeros(1, ctxd)
    pos, neg = [[P, {"pooled_output": z}]], [[N, {"pooled_output": z}]]
    g["P"], g["N"] = P.numpy(), N.numpy()
    gl = torch.randn([2, 4, lat, lat], generator=gen) * 0.5
    with torch.no_grad():
        o = sampling.KSampler().sample(model=mp, seed=5, steps=10, cfg=8.0, denoise=0.45, positive=pos, negative=neg,
                                       latent_image={"samples": gl}, pipeline=True, disable_pbar=True,
                                       sampler_name="euler_ancestral_cfgpp", scheduler="normal")
        o2 = sampling.KSampler().sample(model=mp, seed=6, steps=8, cfg=7.0, denoise=1.0, positive=pos, negative=neg,
                                        latent_image={"samples": torch.zeros(1, 4, lat, lat)}, pipeline=True,
                                        disable_pbar=True, sampler_name="euler_ancestral_cfgpp", scheduler="karras")
    g["anc_latent"] = gl.numpy(); g["anc_img2img"] = o[0]["samples"].numpy(); g["anc_txt2img"] = o2[0]["samples"].numpy()
    # chain: txt2img latents -> LatentUpscale x2 -> 10 steps euler_ancestral_cfgpp / normal, cfg 8, denoise 0.45
    with torch.no_grad():
        base = sampling.KSampler().sample(model=mp, seed=42, steps=6, cfg=7.0, denoise=1.0, positive=pos, negative=neg,
                                          latent_image={"samples": torch.zeros(1, 4, lat, lat)}, pipeline=True,
                                          disable_pbar=True, sampler_name="sample_euler", scheduler="normal",
                                          enable_multiscale=False)
        up = upscale.LatentUpscale().upscale(width=lat * 8 * 2, height=lat * 8 * 2, samples=base[0])
        hi = sampling.KSampler().sample(model=mp, seed=77, steps=10, cfg=8, denoise=0.45, positive=pos, negative=neg,
                                        latent_image=up[0], pipeline=True, disable_pbar=True,
                                        sampler_name="euler_ancestral_cfgpp", scheduler="normal")
    g["hf_base"] = base[0]["samples"].numpy(); g["hf_up"] = up[0]["samples"].numpy(); g["hf_out"] = hi[0]["samples"].numpy()
    np.savez_compressed(os.path.join(OUT, "hires.npz"), **g)
    print("hires.npz", {k: v.shape for k, v in g.items()})


if __name__ == "__main__":
    main()
