"""Golden capture of the Flux arguments AT THE HOOK (cond.py:254-263): the reference's KSampler.sample(flux=True) is run on the
tiny Flux3 with a recording model_function_wrapper; every call's (input, timestep, c_crossattn, y, guidance, cond_or_uncond)
and the result of apply_model are stored so that LdxFluxPatch can be replayed against them on the GPU.
Build container only; writes tests/golden/flux_hook.npz.  See oracle/ref_capture.py / ref_capture_flux_sampler.py (G17)."""
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
    from src.sample import sampling
    from src.BlackForest import Flux
    from src.Model import ModelPatcher
    from src.Device import Device
    from src.cond import cast
    fcfg = ldx.FluxConfig.tiny()
    ucfg = dict(fcfg.reference_kwargs())
    ucfg.update({"image_model": "flux"})
    mc = Flux.Flux(ucfg)
    dev = Device.get_torch_device()
    mc.set_inference_dtype(torch.float32, None)
    mc.custom_operations = cast.manual_cast
    fmodel = mc.get_model({}, "", device=torch.device("cpu"))
    fsd = ldx.weights.synth_state_dict(ldx.weights.flux_state_dict_spec(fcfg), seed=31, dtype=torch.float32)
    fmodel.diffusion_model.load_state_dict(fsd, strict=True)
    fmp = ModelPatcher.ModelPatcher(fmodel, load_device=dev, offload_device=Device.unet_offload_device(), current_device=torch.device("cpu"))
    gen = torch.Generator().manual_seed(11)
    ctx = torch.randn([1, 16, fcfg.context_in_dim], generator=gen)
    y = torch.randn([1, fcfg.vec_in_dim], generator=gen)
    fpos = [[ctx, {"pooled_output": y, "guidance": 3.0}]]
    fneg = [[torch.zeros_like(ctx), {"pooled_output": torch.zeros_like(y), "guidance": 3.0}]]
    rec = []

    def spy(apply_model, params):
        c = params["c"]
        out = apply_model(params["input"], params["timestep"], **c)
        rec.append(dict(input=params["input"].clone(), timestep=params["timestep"].clone(), ctx=c["c_crossattn"].clone(),
                        y=c["y"].clone(), guidance=c["guidance"].clone(), cou=list(params["cond_or_uncond"]), out=out.clone(),
                        keys=sorted(k for k in c.keys())))
        return out

    m = fmp.clone()
    m.set_model_unet_function_wrapper(spy)
    with torch.no_grad():
        o = sampling.KSampler().sample(model=m, seed=9, steps=3, cfg=1, denoise=1, positive=fpos, negative=fneg,
                                       latent_image={"samples": torch.zeros(2, 16, 8, 12)}, pipeline=True, disable_pbar=True,
                                       sampler_name="euler_cfgpp", scheduler="beta", flux=True)
    g = {"n": np.array(len(rec)), "final": o[0]["samples"].numpy(), "c_keys": np.array(",".join(rec[0]["keys"]))}
    for i, r in enumerate(rec):
        for k in ("input", "timestep", "ctx", "y", "guidance", "out"):
            g[f"h{i}_{k}"] = r[k].float().numpy()
        g[f"h{i}_cou"] = np.array(r["cou"])
    np.savez_compressed(os.path.join(ref_capture.OUT, "flux_hook.npz"), **g)
    print("flux_hook.npz", {k: getattr(v, "shape", None) for k, v in g.items()}, rec[0]["keys"])


if __name__ == "__main__":
    main()
