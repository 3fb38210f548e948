"""Golden capture for (G16) the euler_cfgpp sampler with its dy extra steps on the tiny SD1.5 UNet and (G17) the Flux
sampling path — ModelSamplingFlux / CONST, beta scheduler on the 10000-entry shifted table, Flux1 latent format,
euler_cfgpp at cfg 1 with a zeroed negative prompt (pipeline.py:237-262) — through the reference's own KSampler.
Build container only; writes tests/golden/cfgpp.npz.  See oracle/ref_capture.py."""
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
    from src.sample import sampling, ksampler_util
    from src.BlackForest import Flux
    from src.Model import ModelPatcher
    from src.Device import Device
    g = {}

    # ---- G16: euler_cfgpp on the tiny UNet ------------------------------------------------------------
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
    sizes = []

    def spy(apply_model, params):
        sizes.append((params["input"].shape[0], params["input"].shape[-1], float(params["timestep"][0])))
        return apply_model(params["input"], params["timestep"], **params["c"])

    mps = mp.clone()
    mps.set_model_unet_function_wrapper(spy)
    with torch.no_grad():
        o = sampling.KSampler().sample(model=mps, seed=21, steps=8, cfg=7.0, denoise=1.0, positive=pos, negative=neg,
                                       latent_image={"samples": torch.zeros(1, 4, lat, lat)}, pipeline=True, disable_pbar=True,
                                       sampler_name="euler_cfgpp", scheduler="karras")
    g["sd_cfgpp"] = o[0]["samples"].numpy(); g["sd_cfgpp_calls"] = np.array(sizes, dtype=np.float64)
    sizes.clear()
    with torch.no_grad():
        o = sampling.KSampler().sample(model=mps, seed=22, steps=6, cfg=1.0, denoise=1.0, positive=pos, negative=neg,
                                       latent_image={"samples": torch.zeros(2, 4, 18, 14)}, pipeline=True, disable_pbar=True,
                                       sampler_name="euler_cfgpp", scheduler="normal")
    g["sd_cfgpp_cfg1"] = o[0]["samples"].numpy(); g["sd_cfgpp_cfg1_calls"] = np.array(sizes, dtype=np.float64)

    # ---- G17: Flux sampling ------------------------------------------------------------------------------
    fcfg = ldx.FluxConfig.tiny()
    ucfg = dict(fcfg.reference_kwargs())
    ucfg.update({"image_model": "flux"})
    mc = Flux.Flux(ucfg)
    dev = Device.get_torch_device()
    mc.set_inference_dtype(torch.float32, None)
    from src.cond import cast
    mc.custom_operations = cast.manual_cast          # the GGUF loader installs GGMLOps(manual_cast) here; same forward semantics
    fmodel = mc.get_model({}, "", device=torch.device("cpu"))
    fsd = ldx.weights.synth_state_dict(ldx.weights.flux_state_dict_spec(fcfg), seed=31, dtype=torch.float32)
    fmodel.diffusion_model.load_state_dict(fsd, strict=True)
    fmp = ModelPatcher.ModelPatcher(fmodel, load_device=dev, offload_device=Device.unet_offload_device(), current_device=torch.device("cpu"))
    ms = fmodel.model_sampling
    g["flux_sigmas_head"] = ms.sigmas[:8].numpy(); g["flux_sigmas_tail"] = ms.sigmas[-8:].numpy()
    g["flux_sigmas_sample"] = ms.sigmas[::499].numpy()
    for sched in ("beta", "simple"):          # "normal"/"karras" need sigma_min, which ModelSamplingFlux lacks (AttributeError in the reference)
        for steps in (4, 20, 28):
            g[f"flux_{sched}_{steps}"] = ksampler_util.calculate_sigmas(ms, sched, steps).numpy()
    gen = torch.Generator().manual_seed(11)
    lt = 16
    ctx = torch.randn([1, lt, fcfg.context_in_dim], generator=gen)
    y = torch.randn([1, fcfg.vec_in_dim], generator=gen)
    fpos = [[ctx, {"pooled_output": y, "guidance": 3.0}]]
    fneg = [[torch.zeros_like(ctx), {"pooled_output": torch.zeros_like(y), "guidance": 3.0}]]       # ConditioningZeroOut
    g["flux_ctx"], g["flux_y"] = ctx.numpy(), y.numpy()
    calls = []

    def fspy(apply_model, params):
        calls.append((params["input"].shape[0], params["input"].shape[-1], float(params["timestep"][0])))
        return apply_model(params["input"], params["timestep"], **params["c"])

    fmps = fmp.clone()
    fmps.set_model_unet_function_wrapper(fspy)
    with torch.no_grad():
        o = sampling.KSampler().sample(model=fmps, seed=9, steps=6, cfg=1, denoise=1, positive=fpos, negative=fneg,
                                       latent_image={"samples": torch.zeros(1, 16, 8, 12)}, pipeline=True, disable_pbar=True,
                                       sampler_name="euler_cfgpp", scheduler="beta", flux=True)
    g["flux_ks"] = o[0]["samples"].numpy(); g["flux_ks_calls"] = np.array(calls, dtype=np.float64)
    calls.clear()
    gl = torch.randn([2, 16, 8, 8], generator=gen)
    with torch.no_grad():
        o = sampling.KSampler().sample(model=fmps, seed=10, steps=5, cfg=1, denoise=0.6, positive=fpos, negative=fneg,
                                       latent_image={"samples": gl}, pipeline=True, disable_pbar=True,
                                       sampler_name="sample_euler", scheduler="simple", flux=True)
    g["flux_i2i_latent"] = gl.numpy(); g["flux_i2i"] = o[0]["samples"].numpy(); g["flux_i2i_calls"] = np.array(calls, dtype=np.float64)
    # ---- G18: first-block cache (WaveSpeed) on the same sampler runs; threshold chosen so that hits and misses mix
    from src.WaveSpeed import fbcache_nodes, first_block_cache
    hits = []
    orig_similar = first_block_cache.get_can_use_cache

    def spy_can_use(*a, **k):
        r = orig_similar(*a, **k)
        hits.append(int(bool(r)))
        return r

    first_block_cache.get_can_use_cache = spy_can_use
    for thr, tag in ((0.9, "t90"), (0.12, "t12")):
        patched = fbcache_nodes.ApplyFBCacheOnModel().patch((fmp,), "diffusion_model", thr)[0]
        hits.clear()
        with torch.no_grad():
            o = sampling.KSampler().sample(model=patched, seed=9, steps=12, cfg=1, denoise=1, positive=fpos, negative=fneg,
                                           latent_image={"samples": torch.zeros(1, 16, 8, 12)}, pipeline=True, disable_pbar=True,
                                           sampler_name="euler_cfgpp", scheduler="beta", flux=True)
        g[f"fb_{tag}_out"] = o[0]["samples"].numpy(); g[f"fb_{tag}_hits"] = np.array(hits)
        hits.clear()
        with torch.no_grad():
            o = sampling.KSampler().sample(model=patched, seed=10, steps=10, cfg=1, denoise=1, positive=fpos, negative=fneg,
                                           latent_image={"samples": torch.zeros(2, 16, 8, 8)}, pipeline=True, disable_pbar=True,
                                           sampler_name="sample_euler", scheduler="simple", flux=True)
        g[f"fb_{tag}_euler_out"] = o[0]["samples"].numpy(); g[f"fb_{tag}_euler_hits"] = np.array(hits)
        print("fbcache", tag, g[f"fb_{tag}_hits"].tolist(), g[f"fb_{tag}_euler_hits"].tolist())
    first_block_cache.get_can_use_cache = orig_similar
    np.savez_compressed(os.path.join(OUT, "cfgpp.npz"), **g)
    print("cfgpp.npz", {k: v.shape for k, v in g.items()})
    print("sd calls", g["sd_cfgpp_calls"][:, :2].tolist())
    print("flux calls", g["flux_ks_calls"].tolist())
    print("flux i2i calls", g["flux_i2i_calls"].tolist())


if __name__ == "__main__":
    main()
