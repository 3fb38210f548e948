"""LIVE boundary check (build container only): the SHIPPED hook objects inside the REFERENCE's own sampling stack.

Test infrastructure.  tests/test_engine_gpu.py replays RECORDED hook arguments; this script closes the remaining gap: it imports
/root/reference, installs `ldx_amd.hook.LdxUNetPatch` / `LdxFluxPatch` — the classes a user installs — through the reference's own
`ModelPatcher.clone().set_model_unet_function_wrapper` (src/Model/ModelPatcher.py:88-144), lets `ModelPatcher.clone()` deep-copy it and
`model_patches_to()` move it (ModelPatcher.py:108,165-175), and runs the reference's own `KSampler.sample` -> `CFGGuider` ->
`calc_cond_batch` (src/cond/cond.py:150-288) on top of it.  There is no GPU here, so the hook's ENGINE is a CPU test double that answers
with the oracle (oracle/sd15_oracle.py apply_model / flux_apply_model) — everything else on the call path is the shipped code and the
reference's code.  Asserted: the latents equal the un-patched reference run (oracle-vs-reference tolerance), for batch 1 and 3, prompts of
unequal length (77 / 154 tokens -> lcm padding), cfg 7 and cfg 1 (uncond branch skipped: the hook sees batch B, cond_or_uncond [0]).

Usage:  python oracle/ref_check_hook_live.py | tee profiles/r03/ref_check_hook_live.log
"""
import copy
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "oracle"))
sys.path.insert(0, REPO)
import ref_capture  # noqa: E402


class OracleUNetEngine:
    """CPU stand-in for ldx_amd.UNetEngine behind LdxUNetPatch: same attributes the hook touches (.device, .cfg, .denoise)."""

    def __init__(self, cfg, sd, O):
        self.cfg, self.sd, self.O, self.device, self.calls = cfg, {k: v.float() for k, v in sd.items()}, O, torch.device("cpu"), []

    def denoise(self, x, sigma, ctx, out=None):
        self.calls.append((tuple(x.shape), tuple(sigma.shape), tuple(ctx.shape)))
        with torch.no_grad():
            return self.O.apply_model(self.sd, self.cfg, x, sigma, ctx)


class OracleFluxEngine:
    def __init__(self, cfg, sd, O):
        self.cfg, self.sd, self.O, self.device, self.calls = cfg, {k: v.float() for k, v in sd.items()}, O, torch.device("cpu"), []

    def denoise(self, x, sigma, ctx, y, guidance):
        self.calls.append((tuple(x.shape), tuple(ctx.shape)))
        with torch.no_grad():
            return self.O.flux_apply_model(self.sd, self.cfg, x, sigma, ctx, y, guidance)


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def main():
    import ldx_amd as ldx
    from oracle import sd15_oracle as O
    torch.set_num_threads(8)
    ref_capture.enter_reference()
    from src.sample import sampling
    ok = True

    # ------------------------------------------------------------------ SD1.5 UNet hook
    cfg = ldx.UNetConfig.tiny(64, 128)
    sd = ldx.weights.synth_state_dict(ldx.weights.unet_state_dict_spec(cfg), seed=1234)
    model, mp = ref_capture.build_reference_model(cfg, sd)
    gen = torch.Generator().manual_seed(7)
    P = torch.randn([1, 77, 128], generator=gen)
    N = torch.randn([1, 154, 128], generator=gen)              # unequal lengths -> lcm padding by repetition (cond.py:12-52)
    z = torch.zeros(1, 128)
    pos, neg = [[P, {"pooled_output": z}]], [[N, {"pooled_output": z}]]
    eng = OracleUNetEngine(cfg, sd, O)
    patch = ldx.LdxUNetPatch(eng)                               # the shipped class
    mpw = mp.clone()
    mpw.set_model_unet_function_wrapper(patch)
    mpc = mpw.clone()                                           # ModelPatcher.clone deep-copies model_options (ModelPatcher.py:108)
    assert mpc.model_options["model_function_wrapper"] is patch, "deep copy must hand back the same patch object"
    mpc.model_patches_to(torch.device("cpu"))                  # replaces the option with wrapper.to(device) (ModelPatcher.py:165-175)
    assert mpc.model_options["model_function_wrapper"] is patch and copy.deepcopy(patch) is patch
    for B in (1, 3):
        for cfg_scale in (7.0, 1.0):
            for sampler, sched, ms in (("sample_euler", "normal", False), ("dpmpp_2m_cfgpp", "karras", False)):
                kw = dict(seed=11 + B, steps=4, cfg=cfg_scale, sampler_name=sampler, scheduler=sched, denoise=1.0, positive=pos, negative=neg,
                          latent_image={"samples": torch.zeros(B, 4, 16, 16)}, pipeline=True, disable_pbar=True, enable_multiscale=ms)
                with torch.no_grad():
                    want = sampling.KSampler().sample(model=mp, **kw)[0]["samples"]
                    eng.calls.clear()
                    got = sampling.KSampler().sample(model=mpc, **kw)[0]["samples"]
                r = rel(got, want)
                shapes = sorted(set(eng.calls))
                line_ok = r <= 1e-3 and len(eng.calls) > 0 and torch.isfinite(got).all()
                ok = ok and bool(line_ok)
                print(f"UNet hook live: B {B} cfg {cfg_scale} {sampler}/{sched}: rel-L2 vs un-patched reference {r:.2e}  hook calls {len(eng.calls)} "
                      f"shapes {shapes}  {'OK' if line_ok else 'FAIL'}", flush=True)

    # ------------------------------------------------------------------ Flux hook
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
    feng = OracleFluxEngine(fcfg, fsd, O)
    fpatch = ldx.LdxFluxPatch(feng)
    fm = fmp.clone()
    fm.set_model_unet_function_wrapper(fpatch)
    fm = fm.clone()
    fm.model_patches_to(torch.device("cpu"))
    assert fm.model_options["model_function_wrapper"] is fpatch
    for B, hw in ((1, (8, 12)), (2, (8, 8)), (1, (9, 7))):        # the last one: odd latent -> the hook pads circularly like Flux3.forward
        kw = dict(seed=9, steps=3, cfg=1, denoise=1, positive=fpos, negative=fneg, latent_image={"samples": torch.zeros(B, 16, *hw)},
                  pipeline=True, disable_pbar=True, sampler_name="euler_cfgpp", scheduler="beta", flux=True)
        with torch.no_grad():
            want = sampling.KSampler().sample(model=fmp, **kw)[0]["samples"]
            feng.calls.clear()
            got = sampling.KSampler().sample(model=fm, **kw)[0]["samples"]
        r = rel(got, want)
        line_ok = r <= 1e-3 and len(feng.calls) > 0
        ok = ok and bool(line_ok)
        print(f"Flux hook live: B {B} latent {hw}: rel-L2 vs un-patched reference {r:.2e}  hook calls {len(feng.calls)} shapes {sorted(set(feng.calls))}  "
              f"{'OK' if line_ok else 'FAIL'}", flush=True)
    print("ALL OK" if ok else "FAILED")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
