"""LdxUNetPatch — the drop-in object for model_options["model_function_wrapper"].

The reference calls it at src/cond/cond.py:254-263 as
    wrapper(model.apply_model, {"input": x[2B,4,h,w] fp32, "timestep": sigma[2B], "c": {...},
                                "cond_or_uncond": [1, 0]}).chunk(batch_chunks)
and installs it with ModelPatcher.set_model_unet_function_wrapper (src/Model/ModelPatcher.py:138-144).
Same lifecycle contract as the reference's own accelerators, StableFastPatch
(src/StableFast/StableFast.py:230-261) and the FBCache closure (src/WaveSpeed/fbcache_nodes.py:96-111):
callable(model_function, params) -> denoised, `.to(device)` returns self (ModelPatcher.py:165-175
replaces the option with the return value), deep-copy safe (ModelPatcher.clone deep-copies
model_options, ModelPatcher.py:108).
"""
import torch

from .engine import FluxEngine, UNetEngine
from .weights import FluxConfig, UNetConfig


class LdxUNetPatch:
    def __init__(self, engine: UNetEngine):
        self.engine = engine

    @classmethod
    def from_state_dict(cls, state_dict, cfg: UNetConfig = None, device: int = 0, dtype: str = "bf16"):
        """Snapshot weights once (after ModelPatcher.patch_model, i.e. LoRA already merged: SURVEY §8b)."""
        return cls(UNetEngine(cfg or UNetConfig.sd15(), state_dict, device=device, dtype=dtype))

    def __call__(self, model_function, params):
        x = params["input"]
        sigma = params["timestep"]
        c = params["c"]
        ctx = c.get("c_crossattn")
        if ctx is None:
            raise ValueError("LdxUNetPatch: c['c_crossattn'] is required (SD1.5 cross-attention context)")
        for k in ("control", "y"):
            if c.get(k) is not None:
                raise NotImplementedError(f"LdxUNetPatch: conditioning '{k}' is outside the SD1.5 hot path")
        src_device = x.device
        dev = self.engine.device
        cc = c.get("c_concat")            # inpainting UNets (in_channels = 9): ModelBase.py:100-101, concatenated inside the engine's prep kernel
        out = self.engine.denoise(x.to(dev, torch.float32), sigma.to(dev, torch.float32), ctx.to(dev, torch.float32),
                                  c_concat=None if cc is None else cc.to(dev, torch.float32))
        return out if src_device == dev else out.to(src_device)

    def to(self, device):
        return self

    def __deepcopy__(self, memo):
        return self      # the native engine is shared, never duplicated


class LdxFluxPatch:
    """The same hook for the Flux family.  BaseModel.apply_model (src/Model/ModelBase.py:72-133) receives the Flux
    conditioning through the same `c` dict: `c_crossattn` = T5 context [B, Lt, 4096], `y` = pooled CLIP-L vector [B, 768]
    and `guidance` [B] (Flux2.extra_conds, src/BlackForest/Flux.py:781-817); the wrapper returns the CONST-prediction
    denoised latent x - Flux3(x, sigma, ctx, y, guidance) * sigma (src/sample/sampling.py:100-155).  Lifecycle identical to
    LdxUNetPatch (callable(model_function, params), `.to()` returns self, deep-copy safe)."""

    def __init__(self, engine: FluxEngine):
        self.engine = engine

    @classmethod
    def from_state_dict(cls, state_dict, cfg: FluxConfig = None, device: int = 0, dtype: str = "bf16", fp8: bool = False):
        return cls(FluxEngine(cfg or FluxConfig(), state_dict, device=device, dtype=dtype, fp8=fp8))

    def __call__(self, model_function, params):
        x, sigma, c = params["input"], params["timestep"], params["c"]
        ctx, y = c.get("c_crossattn"), c.get("y")
        if ctx is None or y is None:
            raise ValueError("LdxFluxPatch: c['c_crossattn'] (T5 context) and c['y'] (pooled CLIP vector) are required")
        guidance = c.get("guidance")
        if guidance is None and self.engine.cfg.guidance_embed:
            raise ValueError("LdxFluxPatch: this Flux checkpoint embeds guidance; c['guidance'] is required")
        for k in ("c_concat", "control", "attention_mask"):
            if c.get(k) is not None:
                raise NotImplementedError(f"LdxFluxPatch: conditioning '{k}' is outside the Flux hot path")
        src_device = x.device
        dev = self.engine.device
        f = lambda t: None if t is None else t.to(dev, torch.float32)
        h, w = x.shape[-2:]
        xin = f(x)
        if (h | w) & 1:
            # Flux3.forward pads the latent to the 2x2 patch size with CIRCULAR padding and crops the result (Flux.py:749,775-777 /
            # util.pad_to_patch_size); the CONST denoised x - out * sigma is elementwise, so pad -> engine -> crop is the same thing
            xin = torch.nn.functional.pad(xin, (0, w & 1, 0, h & 1), mode="circular")
        out = self.engine.denoise(xin, f(sigma), f(ctx), f(y), f(guidance))
        if (h | w) & 1:
            out = out[:, :, :h, :w].contiguous()
        return out if src_device == dev else out.to(src_device)

    def to(self, device):
        return self

    def __deepcopy__(self, memo):
        return self
