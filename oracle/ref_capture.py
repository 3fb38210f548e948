"""Generate tests/golden/*.npz by IMPORTING THE REFERENCE (/root/reference) in the build container.

Test infrastructure only.  The reference never travels to the GPU box; only the small vectors written here
do (inputs + expected outputs).  Recipe = SURVEY.md Appendix B: scratch cwd with symlinks src/ include/,
a torchsde stub, previews off, SD15.sm_SD15 config, ModelPatcher + sampling.KSampler.

Usage (build container only):  python oracle/ref_capture.py
"""
import os
import sys
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")


def enter_reference():
    scratch = "/tmp/ldx_ref_scratch"
    os.makedirs(scratch, exist_ok=True)
    for name in ("src", "include"):
        link = os.path.join(scratch, name)
        if not os.path.islink(link):
            os.symlink(os.path.join(REF, name), link)
    os.chdir(scratch)
    m = types.ModuleType("torchsde")

    class BrownianTree:  # only symbol referenced (sampling_util.py:202)
        def __init__(self, *a, **k):
            raise RuntimeError("torchsde stub")

    m.BrownianTree = BrownianTree
    sys.modules["torchsde"] = m
    sys.path.insert(0, scratch)
    from src.user import app_instance
    app_instance.app.previewer_var.set(False)


def build_reference_model(cfg, sd):
    from src.SD15 import SD15
    from src.Model import ModelPatcher
    from src.Device import Device
    mc = SD15.sm_SD15(cfg.reference_kwargs())
    dev = Device.get_torch_device()
    mc.set_inference_dtype(torch.float16, Device.unet_manual_cast(torch.float16, dev))
    model = mc.get_model({}, "", device=torch.device("cpu"))
    missing, unexpected = model.diffusion_model.load_state_dict(sd, strict=True), None
    ref_keys = set(model.diffusion_model.state_dict().keys())
    assert ref_keys == set(sd.keys()), (sorted(ref_keys ^ set(sd.keys()))[:10])
    mp = ModelPatcher.ModelPatcher(model, load_device=dev, offload_device=Device.unet_offload_device(),
                                   current_device=torch.device("cpu"))
    return model, mp


def main():
    sys.path.insert(0, REPO)
    import ldx_amd as ldx
    torch.set_num_threads(8)
    enter_reference()
    from src.sample import sampling, ksampler_util
    os.makedirs(OUT, exist_ok=True)
    meta = dict(torch=torch.__version__)

    # ---- G1/G2: schedules and sigma->timestep (a1, a2, a8) ------------------------------------------
    model64, mp64 = build_reference_model(ldx.UNetConfig.tiny(64, 128), ldx.weights.synth_state_dict(
        ldx.weights.unet_state_dict_spec(ldx.UNetConfig.tiny(64, 128)), seed=1234))
    ms = model64.model_sampling
    g = {"sigmas": ms.sigmas.numpy(), "log_sigmas": ms.log_sigmas.numpy()}
    for sched in ("karras", "normal", "simple", "beta"):
        for steps in (1, 8, 20, 28):
            g[f"{sched}_{steps}"] = ksampler_util.calculate_sigmas(ms, sched, steps).numpy()
    for sched in ("karras", "normal"):
        for steps, den in ((10, 0.45), (8, 0.3)):
            ks = sampling.KSampler(model=mp64, steps=steps, sampler="sample_euler", scheduler=sched, denoise=den)
            g[f"{sched}_{steps}_d{den}"] = ks.sigmas.cpu().numpy()
    gen = torch.Generator().manual_seed(5)
    sig_in = torch.cat([ms.sigmas[::37], torch.exp(torch.rand(24, generator=gen) * 6.0 - 3.5),
                        (ms.sigmas[100:108] * ms.sigmas[101:109]).sqrt()])
    g["timestep_in"] = sig_in.numpy()
    g["timestep_out"] = ms.timestep(sig_in).numpy()
    np.savez_compressed(os.path.join(OUT, "schedules.npz"), **g)

    # ---- G6/G7/G8 on tiny configs -------------------------------------------------------------------
    for mcn, ctxd, lat in ((32, 64, 16), (64, 128, 16)):
        cfg = ldx.UNetConfig.tiny(mcn, ctxd)
        sd = ldx.weights.synth_state_dict(ldx.weights.unet_state_dict_spec(cfg), seed=1234)
        model, mp = (model64, mp64) if mcn == 64 else build_reference_model(cfg, sd)
        gen = torch.Generator().manual_seed(7)
        P = torch.randn([1, 77, ctxd], generator=gen)
        N = torch.randn([1, 154, ctxd], generator=gen)          # unequal lengths -> lcm padding by repetition
        z = torch.zeros(1, ctxd)
        pos = [[P, {"pooled_output": z}]]
        neg = [[N, {"pooled_output": z}]]
        g = {"P": P.numpy(), "N": N.numpy()}

        # G6: raw UNet forward + apply_model
        xg = torch.randn([2, 4, lat, lat], generator=gen)
        tt = torch.tensor([981.0, 40.0])
        ctx = torch.randn([2, 77, ctxd], generator=gen)
        with torch.no_grad():
            y = model.diffusion_model(xg, tt, context=ctx, transformer_options={})
            sig = torch.tensor([7.5, 0.3])
            den = model.apply_model(xg, sig, c_crossattn=ctx, transformer_options={})
            # odd latent size exercises Downsample ceil / Upsample-to-skip-shape (ResBlock.py:120-136)
            xo = torch.randn([1, 4, 18, 12], generator=gen)
            yo = model.diffusion_model(xo, torch.tensor([500.0]), context=ctx[:1], transformer_options={})
        g.update(unet_x=xg.numpy(), unet_t=tt.numpy(), unet_ctx=ctx.numpy(), unet_y=y.float().numpy(),
                 am_sigma=sig.numpy(), am_out=den.numpy(), odd_x=xo.numpy(), odd_y=yo.float().numpy())

        # G7: the wrapper contract, recorded at the hook (cond.py:254-263)
        rec = []

        def wrapper(apply_model, params):
            out = apply_model(params["input"], params["timestep"], **params["c"])
            rec.append(dict(input=params["input"].clone(), timestep=params["timestep"].clone(),
                            ctx=params["c"]["c_crossattn"].clone(), cou=list(params["cond_or_uncond"]), out=out.clone()))
            return out

        mpw = mp.clone()
        mpw.set_model_unet_function_wrapper(wrapper)
        lat3 = torch.zeros(3, 4, 8, 8)
        with torch.no_grad():
            out = sampling.KSampler().sample(model=mpw, seed=11, steps=2, cfg=7.0, sampler_name="sample_euler",
                                             scheduler="karras", denoise=1.0, positive=pos, negative=neg,
                                             latent_image={"samples": lat3}, pipeline=True, disable_pbar=True,
                                             enable_multiscale=False)
        for i, r in enumerate(rec):
            g[f"hook{i}_input"] = r["input"].numpy(); g[f"hook{i}_timestep"] = r["timestep"].numpy()
            g[f"hook{i}_ctx"] = r["ctx"].numpy(); g[f"hook{i}_cou"] = np.array(r["cou"]); g[f"hook{i}_out"] = r["out"].numpy()
        g["hook_n"] = np.array(len(rec))
        g["hook_final"] = out[0]["samples"].numpy()

        # G8: end-to-end KSampler.sample latents (a3-a8)
        runs = {
            "euler_ms_off": dict(sampler_name="sample_euler", scheduler="normal", enable_multiscale=False),
            "euler_ms_on": dict(sampler_name="sample_euler", scheduler="normal", enable_multiscale=True),
            "euler_forced": dict(sampler_name="euler", scheduler="karras", enable_multiscale=False),
            "dpmpp2m": dict(sampler_name="dpmpp_2m_cfgpp", scheduler="karras", enable_multiscale=False),
        }
        for name, kw in runs.items():
            sizes = []

            def spy(apply_model, params):
                sizes.append(params["input"].shape[-1])
                return apply_model(params["input"], params["timestep"], **params["c"])

            mps = mp.clone()
            mps.set_model_unet_function_wrapper(spy)
            with torch.no_grad():
                o = sampling.KSampler().sample(model=mps, seed=42, steps=20, cfg=7.0, denoise=1.0, positive=pos,
                                               negative=neg, latent_image={"samples": torch.zeros(1, 4, lat, lat)},
                                               pipeline=True, disable_pbar=True, **kw)
            g[f"ks_{name}"] = o[0]["samples"].numpy()
            g[f"ks_{name}_res"] = np.array(sizes)
        # img2img-style: non-empty latent + denoise < 1
        gl = torch.randn([1, 4, lat, lat], generator=gen) * 0.5
        with torch.no_grad():
            o = sampling.KSampler().sample(model=mp, seed=3, steps=10, cfg=5.0, denoise=0.45, positive=pos, negative=neg,
                                           latent_image={"samples": gl}, pipeline=True, disable_pbar=True,
                                           sampler_name="sample_euler", scheduler="normal", enable_multiscale=False)
        g["ks_img2img_latent"] = gl.numpy()
        g["ks_img2img"] = o[0]["samples"].numpy()
        np.savez_compressed(os.path.join(OUT, f"unet_mc{mcn}.npz"), **g)
        print("wrote", f"unet_mc{mcn}.npz", {k: v.shape for k, v in g.items() if k.startswith("ks_")})
    with open(os.path.join(OUT, "META.txt"), "w") as f:
        f.write(f"generated by oracle/ref_capture.py from /root/reference @ 2025-07-25 snapshot; torch {meta['torch']}\n"
                "weights: ldx.weights.synth_state_dict(spec, seed=1234) (fp16 storage), fp32 compute (manual_cast)\n")


if __name__ == "__main__":
    main()
