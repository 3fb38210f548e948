"""UNetEngine — Python handle on the native SD1.5 UNet engine (libldx.so).

Host-side responsibilities only: hand the state dict to the C ABI once, build the two lookup tables the
reference builds on the host (sigma table and sinusoidal timestep embeddings), and pass device pointers.
Reference counterparts: BaseModel.__init__/load_model_weights/apply_model (src/Model/ModelBase.py:38-202).
"""
import ctypes as C
import math

import numbers

import torch

from . import lib
from .weights import T5Config  # noqa: F401
from .weights import UNetConfig, VAEConfig, CLIPConfig, FluxConfig


def sd15_sigmas():
    """ModelSamplingDiscrete._register_schedule + set_sigmas (src/sample/sampling.py:224-289) with
    make_beta_schedule("linear", 1000, 0.00085, 0.012) (src/sample/sampling_util.py:18-39).
    Returns (sigmas fp32 [1000], log_sigmas fp32 [1000]); fp64 until the final cast, like the reference."""
    betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=torch.float64) ** 2
    alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
    sigmas = ((1 - alphas_cumprod) / alphas_cumprod) ** 0.5
    return sigmas.float(), sigmas.log().float()


def timestep_index(log_sigmas: torch.Tensor, sigma) -> torch.Tensor:
    """ModelSamplingDiscrete.timestep (src/sample/sampling.py:309-320, called from BaseModel.apply_model, ModelBase.py:112) on the HOST: the reference's
    expression `dists = sigma.log() - log_sigmas[:, None]; dists.abs().argmin(dim=0)`.  Integer work, so it has to be exact — and one ulp of log(sigma)
    decides the index at a near-tie (sigma at the geometric midpoint of two table entries: the `normal` scheduler's fractional timesteps).
    sigma: float / host tensor; returns int64 [n]."""
    sg = torch.as_tensor(sigma, dtype=torch.float32, device="cpu").reshape(-1)
    # the logarithm CORRECTLY ROUNDED (fp64 log, rounded once): torch's fp32 CPU log is correctly rounded for 99.98 % of inputs, and where it is not its last
    # bit follows the host's vector ISA — the GPU box's host put the golden near-tie sigma 0.3473117 (tests/golden/schedules.npz, captured in the build
    # container: index 101) on the other side.  The correctly rounded value reproduces every golden index on every host, and the device lookup
    # (csrc/misc.hip nearest_log_sigma) computes the same function.
    log_sigma = sg.double().log().float()
    return (log_sigma - log_sigmas[:, None]).abs().argmin(dim=0)


def timestep_embedding_table(n: int, dim: int, max_period: int = 10000) -> torch.Tensor:
    """timestep_embedding (src/sample/sampling_util.py:56-76) evaluated at t = 0..n-1 -> [n][dim] fp32."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(start=0, end=half, dtype=torch.float32) / half)
    args = torch.arange(n)[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1).contiguous()


class UNetEngine:
    def __init__(self, cfg: UNetConfig, state_dict, device: int = 0, dtype: str = "bf16", graph: bool = False):
        self._lib = lib.load()
        self._h = C.c_void_p()
        self.cfg = cfg
        self.device = torch.device("cuda", device)
        self.dtype = dtype
        c = lib.ldx_unet_config()
        c.compute_dtype = {"bf16": lib.LDX_BF16, "f16": lib.LDX_F16, "fp16": lib.LDX_F16}[dtype]
        c.in_channels, c.out_channels, c.model_channels = cfg.in_channels, cfg.out_channels, cfg.model_channels
        c.num_levels = len(cfg.channel_mult)
        for i, v in enumerate(cfg.channel_mult):
            c.channel_mult[i] = v
        for i, v in enumerate(cfg.num_res_blocks):
            c.num_res_blocks[i] = v
        for i, v in enumerate(cfg.transformer_depth):
            c.transformer_depth[i] = v
        for i, v in enumerate(cfg.transformer_depth_output):
            c.transformer_depth_output[i] = v
        c.transformer_depth_middle = cfg.transformer_depth_middle
        c.num_heads, c.context_dim = cfg.num_heads, cfg.context_dim
        lib.check(self._lib.ldx_create(C.byref(c), device, C.byref(self._h)), "ldx_create")
        prefix = "model.diffusion_model."
        for k, t in state_dict.items():
            if k.startswith(prefix):
                k = k[len(prefix):]
            t = t.detach().to("cpu").contiguous()
            if t.dtype not in (torch.float16, torch.bfloat16, torch.float32):
                t = t.float()
            shape = (C.c_int64 * t.dim())(*t.shape)
            lib.check(self._lib.ldx_load_tensor(self._h, k.encode(), lib.ptr(t), lib.torch_dtype_code(t.dtype),
                                                shape, t.dim()), f"ldx_load_tensor({k})")
        self.sigmas, self.log_sigmas = sd15_sigmas()
        temb = timestep_embedding_table(self.sigmas.numel(), cfg.model_channels)
        lib.check(self._lib.ldx_set_tables(self._h, lib.ptr(self.log_sigmas), self.log_sigmas.numel(),
                                           lib.ptr(temb), temb.shape[1]), "ldx_set_tables")
        lib.check(self._lib.ldx_finalize(self._h), "ldx_finalize")
        if graph:
            self.set_graph_mode(True)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.ldx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_graph_mode(self, on: bool):
        lib.check(self._lib.ldx_set_graph_mode(self._h, int(on)), "ldx_set_graph_mode")

    def set_context_cache(self, on: bool = True):
        """ldx_unet_context_cache: promise that a ctx buffer's contents stay put until the next call of this method (which invalidates the cache)."""
        lib.check(self._lib.ldx_unet_context_cache(self._h, int(on)), "ldx_unet_context_cache")
        self._ctx_cached = bool(on)

    def invalidate_context(self):
        """A cached context buffer was rewritten in place: drop its projections (the mode stays as it is)."""
        self.set_context_cache(getattr(self, "_ctx_cached", False))

    def _ctx_mode(self, cached: bool):
        # every entry point states whether ITS ctx may be cached; the default is the reference's behaviour (recompute): a caller that hands over a
        # fresh tensor per step (the hook) may get the address of the previous one back from the allocator, with other contents
        if bool(cached) != getattr(self, "_ctx_cached", False):
            self.set_context_cache(cached)

    def graph_stats(self):
        """(captures, replays) of the engine's hipGraph path since it was created."""
        c, r = C.c_int64(0), C.c_int64(0)
        lib.check(self._lib.ldx_graph_stats(self._h, C.byref(c), C.byref(r)), "ldx_graph_stats")
        return int(c.value), int(r.value)

    def timestep_index(self, sigma):
        """ModelSamplingDiscrete.timestep (sample/sampling.py:309-320) with the reference's own torch expression on the HOST: integer work, so it has to be
        bit-exact, and a device logf may differ from the host's log in the last bit (near-ties pick the other index).  sigma: host tensor / float."""
        return timestep_index(self.log_sigmas, sigma)

    def timestep_device(self, sigma):
        """The device's own lookup (ldx_unet_timestep): what ldx_unet_denoise runs when sigma lives on the GPU.  Returns int32 [n] on the device."""
        sg = sigma.to(device=self.device, dtype=torch.float32).contiguous().reshape(-1)
        out = torch.empty(sg.numel(), dtype=torch.int32, device=sg.device)
        lib.check(self._lib.ldx_unet_timestep(self._h, lib.ptr(sg), sg.numel(), lib.ptr(out), lib.current_stream_ptr()), "ldx_unet_timestep")
        return out

    def _ctx_arg(self, ctx, device, ctx_cached):
        """ctx as the C ABI wants it.  With the context cache on, the engine keys the cached k|v projections on the POINTER: a conversion here would hand it a
        temporary whose address the allocator gives to the next temporary too (other contents, same key) - so a cached ctx must already be in place."""
        if ctx_cached:
            assert ctx.is_cuda and ctx.dtype == torch.float32 and ctx.is_contiguous(), "ctx_cached=True needs a contiguous CUDA fp32 ctx (the cache is keyed on its address)"
            return ctx
        return ctx.to(device=device, dtype=torch.float32).contiguous()

    def _run(self, fn, x, s, ctx, out, ctx_cached=False, t_idx=None):
        self._ctx_mode(ctx_cached)
        assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 4, "x must be a CUDA fp32 NCHW tensor"
        b2, ch, h, w = x.shape
        assert ch == self.cfg.in_channels
        x = x.contiguous()
        s = s.to(device=x.device, dtype=torch.float32).contiguous()
        ctx = self._ctx_arg(ctx, x.device, ctx_cached)
        assert s.numel() == b2 and ctx.dim() == 3 and ctx.shape[0] == b2 and ctx.shape[2] == self.cfg.context_dim
        if out is None:
            out = torch.empty((b2, self.cfg.out_channels, h, w), device=x.device, dtype=torch.float32)
        if t_idx is not None:
            t = t_idx.to(device=x.device, dtype=torch.float32).contiguous()
            assert t.numel() == b2
            lib.check(self._lib.ldx_unet_denoise_t(self._h, lib.ptr(x), lib.ptr(s), lib.ptr(t), lib.ptr(ctx), b2, h, w, ctx.shape[1], lib.ptr(out),
                                                   lib.current_stream_ptr()), "ldx_unet_denoise_t")
            return out
        lib.check(fn(self._h, lib.ptr(x), lib.ptr(s), lib.ptr(ctx), b2, h, w, ctx.shape[1], lib.ptr(out),
                     lib.current_stream_ptr()), fn.__name__)
        return out

    def denoise(self, x, sigma, ctx, out=None, c_concat=None, ctx_cached=False):
        """BaseModel.apply_model (ModelBase.py:72-133): x fp32 [B2,4,h,w], sigma [B2] (values), ctx [B2,M,768].
        c_concat [B2,in_channels-4,h,w] (inpainting UNets, ModelBase.py:100-101): appended unscaled behind the scaled x inside the engine's prep kernel."""
        if c_concat is None:
            # sigma on the host (the reference's CPU path, the hook's recorded calls): the timestep index comes from the reference's own torch expression
            t_idx = self.timestep_index(sigma) if (torch.is_tensor(sigma) and not sigma.is_cuda) else None
            return self._run(self._lib.ldx_unet_denoise, x, sigma, ctx, out, ctx_cached, t_idx=t_idx)
        self._ctx_mode(ctx_cached)
        assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 4, "x must be a CUDA fp32 NCHW tensor"
        b2, ch, h, w = x.shape
        cc = c_concat.to(device=x.device, dtype=torch.float32).contiguous()
        assert cc.dim() == 4 and cc.shape[0] == b2 and cc.shape[2:] == x.shape[2:] and ch + cc.shape[1] == self.cfg.in_channels, "c_concat must be [B2, in_channels - C(x), h, w]"
        x = x.contiguous()
        s = sigma.to(device=x.device, dtype=torch.float32).contiguous()
        ctx = self._ctx_arg(ctx, x.device, ctx_cached)
        assert s.numel() == b2 and ctx.dim() == 3 and ctx.shape[0] == b2 and ctx.shape[2] == self.cfg.context_dim
        if out is None:
            out = torch.empty((b2, self.cfg.out_channels, h, w), device=x.device, dtype=torch.float32)
        lib.check(self._lib.ldx_unet_denoise_concat(self._h, lib.ptr(x), lib.ptr(s), lib.ptr(ctx), lib.ptr(cc), cc.shape[1], b2, h, w, ctx.shape[1],
                                                    lib.ptr(out), lib.current_stream_ptr()), "ldx_unet_denoise_concat")
        return out

    def denoise_cfg(self, x, sigma: float, ctx, out=None, ctx_cached=False):
        """One CFG evaluation (calc_cond_batch, cond.py:186-226): x fp32 [B,4,h,w] is read by both halves of the [uncond x B; cond x B]
        batch, sigma is one scalar, ctx [2B,M,768]; returns [2B,4,h,w].  No torch kernel runs: the broadcast of x and sigma happens in
        the engine's own boundary kernels (ldx_unet_denoise_cfg)."""
        self._ctx_mode(ctx_cached)
        assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous(), "x must be a contiguous CUDA fp32 NCHW tensor"
        b, ch, h, w = x.shape
        assert ch == self.cfg.in_channels
        assert ctx.is_cuda and ctx.dtype == torch.float32 and ctx.is_contiguous() and ctx.dim() == 3 and ctx.shape[0] == 2 * b and ctx.shape[2] == self.cfg.context_dim
        if out is None:
            out = torch.empty((2 * b, self.cfg.out_channels, h, w), device=x.device, dtype=torch.float32)
        # sigma is a host scalar here: its timestep index is computed with the reference's own torch ops (bit-exact by construction, see timestep_index)
        t_index = int(self.timestep_index(float(sigma))[0])
        lib.check(self._lib.ldx_unet_denoise_cfg_t(self._h, lib.ptr(x), float(sigma), t_index, lib.ptr(ctx), b, h, w, ctx.shape[1], lib.ptr(out),
                                                   lib.current_stream_ptr()), "ldx_unet_denoise_cfg_t")
        return out

    def forward(self, x, timesteps, ctx, out=None):
        """UNetModel1.forward (unet.py:679-770): integer timesteps (as floats), unscaled input."""
        return self._run(self._lib.ldx_unet_forward, x, timesteps, ctx, out)

    def profile(self, on: bool, reset: bool = True):
        lib.check(self._lib.ldx_profile(self._h, int(on), int(reset)), "ldx_profile")

    def profile_report(self) -> dict:
        import json
        buf = C.create_string_buffer(1 << 18)
        lib.check(self._lib.ldx_profile_report(self._h, buf, len(buf)), "ldx_profile_report")
        return json.loads(buf.value.decode())

    def plan_info(self):
        """launches / algorithmic flops (the reference's arithmetic for this evaluation) / arena bytes of the current plan, plus the flops as EXECUTED
        (`flops_executed`; smaller than `flops` by `flops_shared` when the plan computes the shared CFG prefix once, ldx_unet_cfg_share)."""
        n, f, a = C.c_int64(), C.c_double(), C.c_int64()
        lib.check(self._lib.ldx_plan_info(self._h, C.byref(n), C.byref(f), C.byref(a)), "ldx_plan_info")
        ex, sh = C.c_double(), C.c_double()
        lib.check(self._lib.ldx_plan_flops(self._h, C.byref(ex), C.byref(sh)), "ldx_plan_flops")
        return {"launches": n.value, "flops": f.value, "arena_bytes": a.value, "flops_executed": ex.value, "flops_shared": sh.value}

    def set_cfg_share(self, mode=True):
        """ldx_unet_cfg_share: denoise_cfg computes the part of the UNet in front of the first cross-attention once for both CFG halves.
        False / 0: never; True / 1 (default): where it pays (>= 8192 rows per half); 2: whenever possible (tests on small latents)."""
        lib.check(self._lib.ldx_unet_cfg_share(self._h, int(mode)), "ldx_unet_cfg_share")


def _load_state_dict(L, h, state_dict, strip=()):
    for k, t in state_dict.items():
        for pre in strip:
            if k.startswith(pre):
                k = k[len(pre):]
        t = t.detach().to("cpu").contiguous()
        if t.dtype not in (torch.float16, torch.bfloat16, torch.float32):
            t = t.float()
        shape = (C.c_int64 * t.dim())(*t.shape)
        lib.check(L.ldx_load_tensor(h, k.encode(), lib.ptr(t), lib.torch_dtype_code(t.dtype), shape, t.dim()),
                  f"ldx_load_tensor({k})")


class VAEDecoderEngine:
    """VAE.decode (src/AutoEncoders/VariationalAE.py:690-722): latent (already / 0.18215) -> NHWC fp32 in [0, 1]."""

    def __init__(self, cfg: VAEConfig, state_dict, device: int = 0, dtype: str = "bf16"):
        self._lib = lib.load()
        self._h = C.c_void_p()
        self.cfg, self.device = cfg, torch.device("cuda", device)
        c = lib.ldx_vae_config()
        c.compute_dtype = {"bf16": lib.LDX_BF16, "f16": lib.LDX_F16, "fp16": lib.LDX_F16}[dtype]
        c.z_channels, c.ch, c.num_levels = cfg.z_channels, cfg.ch, len(cfg.ch_mult)
        for i, v in enumerate(cfg.ch_mult):
            c.ch_mult[i] = v
        c.num_res_blocks, c.out_ch, c.use_post_quant = cfg.num_res_blocks, cfg.out_ch, int(cfg.use_post_quant)
        lib.check(self._lib.ldx_vae_create(C.byref(c), device, C.byref(self._h)), "ldx_vae_create")
        _load_state_dict(self._lib, self._h, state_dict, strip=("first_stage_model.",))
        lib.check(self._lib.ldx_finalize(self._h), "ldx_finalize")

    def decode(self, z):
        assert z.is_cuda and z.dtype == torch.float32 and z.dim() == 4
        b, _, h, w = z.shape
        out = torch.empty((b, 8 * h, 8 * w, self.cfg.out_ch), device=z.device, dtype=torch.float32)
        lib.check(self._lib.ldx_vae_decode(self._h, lib.ptr(z.contiguous()), b, h, w, lib.ptr(out), lib.current_stream_ptr()),
                  "ldx_vae_decode")
        return out

    def encode_moments(self, pixels):
        """Deterministic part of VAE.encode (VariationalAE.py:725-760): pixels [B,H,W,3] fp32 in [0,1] ->
        moments [B, 2*z, H/8, W/8] (mean | logvar).  Needs encoder.* / quant_conv.* in the state dict."""
        assert pixels.is_cuda and pixels.dtype == torch.float32 and pixels.dim() == 4 and pixels.shape[-1] == 3
        b, h, w, _ = pixels.shape
        f = 2 ** (len(self.cfg.ch_mult) - 1)
        out = torch.empty((b, 2 * self.cfg.z_channels, h // f, w // f), device=pixels.device, dtype=torch.float32)
        lib.check(self._lib.ldx_vae_encode(self._h, lib.ptr(pixels.contiguous()), b, h, w, lib.ptr(out), lib.current_stream_ptr()),
                  "ldx_vae_encode")
        return out

    def encode(self, pixels):
        """VAE.encode (VariationalAE.py:725-760).  vae_encode_crop_pixels (:677-688) computes the cropped sizes and
        throws them away, so nothing is cropped: every Downsample floors.  Then DiagonalGaussianDistribution.sample
        (:42-51) — torch.randn(mean.shape) from the *global CPU RNG* exactly as the reference draws it, uploaded."""
        mom = self.encode_moments(pixels[..., :3].to(self.device, torch.float32))
        mean, logvar = torch.chunk(mom, 2, dim=1)
        eps = torch.randn(mean.shape).to(self.device)
        return mean + torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0)) * eps

    profile = UNetEngine.profile
    profile_report = UNetEngine.profile_report
    plan_info = UNetEngine.plan_info
    close = UNetEngine.close
    __del__ = UNetEngine.__del__


def _bislerp_data(length_old: int, length_new: int):
    """generate_bilinear_data (src/Utilities/upscale.py:61-97): the reference builds the index / ratio arrays with
    F.interpolate(mode="bilinear") of an arange; reproduced verbatim on the host (length_new floats)."""
    c1 = torch.arange(length_old, dtype=torch.float32).reshape(1, 1, 1, -1)
    c1 = torch.nn.functional.interpolate(c1, size=(1, length_new), mode="bilinear")
    ratios = (c1 - c1.floor()).reshape(-1)
    c2 = torch.arange(length_old, dtype=torch.float32).reshape(1, 1, 1, -1) + 1
    c2[:, :, :, -1] -= 1
    c2 = torch.nn.functional.interpolate(c2, size=(1, length_new), mode="bilinear")
    return ratios.contiguous(), c1.reshape(-1).to(torch.int32), c2.reshape(-1).to(torch.int32)


def bislerp(samples, width: int, height: int):
    """bislerp (src/Utilities/upscale.py:5-128) = LatentUpscale.upscale's resampler (HiresFix, pipeline.py:346-350):
    a slerp pass along W then one along H, each one ldx_bislerp_pass launch on fp32 NCHW device tensors."""
    assert samples.is_cuda and samples.dim() == 4
    L = lib.load()
    x = samples.float().contiguous()
    n, c, h, w = x.shape
    dev = x.device
    r, c1, c2 = (t.to(dev) for t in _bislerp_data(w, width))
    y = torch.empty((n, c, h, width), device=dev, dtype=torch.float32)
    lib.check(L.ldx_bislerp_pass(lib.ptr(x), lib.ptr(y), n, c, h, w, 1, width, lib.ptr(c1), lib.ptr(c2), lib.ptr(r),
                                 lib.current_stream_ptr()), "ldx_bislerp_pass")
    r2, d1, d2 = (t.to(dev) for t in _bislerp_data(h, height))
    z = torch.empty((n, c, height, width), device=dev, dtype=torch.float32)
    lib.check(L.ldx_bislerp_pass(lib.ptr(y), lib.ptr(z), n, c, h, width, 0, height, lib.ptr(d1), lib.ptr(d2), lib.ptr(r2),
                                 lib.current_stream_ptr()), "ldx_bislerp_pass")
    return z.to(samples.dtype)


def latent_upscale(samples, width: int, height: int):
    """LatentUpscale.upscale (upscale.py:149-166): target latent size = pixel size // 8."""
    return bislerp(samples, width // 8, height // 8)


class CLIPTextEngine:
    """CLIPTextModel_.forward (src/clip/CLIPTextModel.py:51-107) + the token-weight logic of
    ClipTokenWeightEncoder.encode_token_weights (src/SD15/SDClip.py:36-97) on the host."""

    def __init__(self, cfg: CLIPConfig, state_dict, device: int = 0, dtype: str = "bf16"):
        self._lib = lib.load()
        self._h = C.c_void_p()
        self.cfg, self.device = cfg, torch.device("cuda", device)
        c = lib.ldx_clip_config()
        c.compute_dtype = {"bf16": lib.LDX_BF16, "f16": lib.LDX_F16, "fp16": lib.LDX_F16}[dtype]
        c.hidden_size, c.num_layers, c.num_heads = cfg.hidden_size, cfg.num_layers, cfg.num_heads
        c.intermediate_size, c.max_positions, c.vocab_size = cfg.intermediate_size, cfg.max_positions, cfg.vocab_size
        lib.check(self._lib.ldx_clip_create(C.byref(c), device, C.byref(self._h)), "ldx_clip_create")
        _load_state_dict(self._lib, self._h, state_dict, strip=("text_model.",))       # incl. the optional "text_projection.weight"
        lib.check(self._lib.ldx_finalize(self._h), "ldx_finalize")
        self._tok_dtype = next(v.dtype for k, v in state_dict.items() if k.endswith("embeddings.token_embedding.weight"))

    def forward(self, tokens, intermediate_output=None):
        """tokens: int tensor [B][T] -> (last [B,T,E] after final LN, intermediate (final-LN'd) or None, pooled)."""
        ids = tokens.to(self.device, torch.int32).contiguous()
        b, t = ids.shape
        last = torch.empty((b, t, self.cfg.hidden_size), device=self.device, dtype=torch.float32)
        inter = torch.empty_like(last) if intermediate_output is not None else None
        lib.check(self._lib.ldx_clip_encode(self._h, lib.ptr(ids), b, t, int(intermediate_output or 0), lib.ptr(last),
                                            lib.ptr(inter), lib.current_stream_ptr()), "ldx_clip_encode")
        # pooled output: row at argmax(tokens == eos_token_id) (CLIPTextModel.py:98-106; eos id 2 -> position 0 quirk), then the optional
        # text_projection (CLIPTextModel.py:152-163) — both inside the engine (ldx_clip_pooled), no torch kernel
        pooled = torch.empty((b, self.cfg.hidden_size), device=self.device, dtype=torch.float32)
        lib.check(self._lib.ldx_clip_pooled(self._h, lib.ptr(last), lib.ptr(ids), b, t, int(self.cfg.eos_token_id), lib.ptr(pooled),
                                            lib.current_stream_ptr()), "ldx_clip_pooled")
        return last, inter, pooled

    _extra_rows = 0

    def _textual_embeddings(self, chunks, pad_token):
        """SDClipModel.set_up_textual_embeddings (SDClip.py:213-267): a token that is a vector of the model's width becomes
        the id vocab_size + k of an appended table row; a vector of another width is dropped and the chunk re-padded."""
        e, nxt, rows, out = self.cfg.hidden_size, self.cfg.vocab_size, [], []
        for chunk in chunks:
            ids = []
            for t in chunk:
                if isinstance(t, numbers.Integral):
                    ids.append(int(t))
                elif t.shape[0] == e:
                    rows.append(torch.as_tensor(t)); ids.append(nxt); nxt += 1
            ids += [pad_token] * (len(chunk) - len(ids))
            out.append(ids)
        return out, rows

    def encode_token_weights(self, token_weight_pairs, layer_idx=-2, special_tokens=(49406, 49407, 49407)):
        """ClipTokenWeightEncoder.encode_token_weights (SDClip.py:36-97) for a list of 77-token (id, weight) chunks; an id
        may be a textual-inversion vector (prompt.tokenize_with_weights with embeddings=...)."""
        to_encode, has_weights, max_len = [], False, 0
        for x in token_weight_pairs:
            toks = [a[0] for a in x]
            max_len = max(max_len, len(toks))
            has_weights = has_weights or not all(a[1] == 1.0 for a in x)
            to_encode.append(toks)
        sections = len(to_encode)
        if has_weights or sections == 0:
            start, end, pad = special_tokens
            to_encode.append([start, end] + [pad] * (max_len - 2))           # gen_empty_tokens (SDClip.py:10-20)
        to_encode, extra = self._textual_embeddings(to_encode, special_tokens[2])
        if extra or self._extra_rows:
            # the reference copies the vectors into a table of the stored weights' dtype (SDClip.py:249-258): same rounding here
            rows = torch.stack(extra).to(self._tok_dtype).float().contiguous() if extra else None
            lib.check(self._lib.ldx_clip_set_extra_embeddings(self._h, None if rows is None else C.c_void_p(rows.data_ptr()), len(extra)),
                      "ldx_clip_set_extra_embeddings")
            self._extra_rows = len(extra)
        last, inter, pooled = self.forward(torch.tensor(to_encode, dtype=torch.int64), intermediate_output=layer_idx)
        out = (inter if layer_idx is not None else last).float().cpu()
        output = []
        for k in range(sections):
            z = out[k:k + 1].clone()
            if has_weights:
                z_empty = out[-1]
                for j in range(z.shape[1]):
                    wgt = token_weight_pairs[k][j][1]
                    if wgt != 1.0:
                        z[0][j] = (z[0][j] - z_empty[j]) * wgt + z_empty[j]
            output.append(z)
        cond = out[-1:] if not output else torch.cat(output, dim=-2)
        return cond, pooled[0:1].float().cpu()

    close = UNetEngine.close
    __del__ = UNetEngine.__del__


class ESRGANEngine:
    """RRDBNet.forward (src/UltimateSDUpscale/RDRB.py:216-471) + tiled_scale (src/Utilities/util.py:406-640) as
    UpscaleModelLoader / ImageUpscaleWithModel use them (USDU_upscaler.py:48-96)."""

    def __init__(self, cfg, state_dict, device: int = 0, dtype: str = "bf16"):
        import math as _m
        from .weights import esrgan_new_to_old_arch
        self._lib = lib.load()
        self._h = C.c_void_p()
        self.cfg, self.device = cfg, torch.device("cuda", device)
        c = lib.ldx_esrgan_config()
        c.compute_dtype = {"bf16": lib.LDX_BF16, "f16": lib.LDX_F16, "fp16": lib.LDX_F16}[dtype]
        c.in_nc, c.out_nc, c.nf, c.gc, c.num_blocks, c.num_upscale = cfg.in_nc, cfg.out_nc, cfg.nf, cfg.gc, cfg.num_blocks, int(_m.log2(cfg.scale))
        lib.check(self._lib.ldx_esrgan_create(C.byref(c), device, C.byref(self._h)), "ldx_esrgan_create")
        _load_state_dict(self._lib, self._h, esrgan_new_to_old_arch(state_dict))
        lib.check(self._lib.ldx_finalize(self._h), "ldx_finalize")

    def forward(self, pixels):
        """pixels [B,H,W,in_nc] fp32 (device) -> [B, s*H, s*W, out_nc] fp32."""
        assert pixels.is_cuda and pixels.dtype == torch.float32 and pixels.dim() == 4
        b, h, w, _ = pixels.shape
        s = self.cfg.scale
        out = torch.empty((b, s * h, s * w, self.cfg.out_nc), device=pixels.device, dtype=torch.float32)
        lib.check(self._lib.ldx_esrgan_forward(self._h, lib.ptr(pixels.contiguous()), b, h, w, lib.ptr(out), lib.current_stream_ptr()),
                  "ldx_esrgan_forward")
        return out

    def upscale(self, image, tile: int = 512, overlap: int = 32):
        """ImageUpscaleWithModel.upscale (USDU_upscaler.py:48-96): tiled_scale(tile 512, overlap 32) then clamp to [0, 1].
        image [B,H,W,3] fp32 in [0,1] (any device) -> [B, sH, sW, 3] on the engine's device.  Tile positions, the
        single-tile shortcut and the feather mask follow tiled_scale_multidim (util.py:406-600) exactly."""
        L, s = self._lib, self.cfg.scale
        image = image.to(self.device, torch.float32).contiguous()
        b, h, w, c = image.shape
        outs = []
        for bi in range(b):
            img = image[bi:bi + 1]
            if h <= tile and w <= tile:
                out = self.forward(img)
                lib.check(L.ldx_tile_finish(lib.ptr(out), None, out.numel(), 1, lib.current_stream_ptr()), "ldx_tile_finish")
                outs.append(out)
                continue
            out = torch.zeros((1, s * h, s * w, self.cfg.out_nc), device=self.device, dtype=torch.float32)
            div = torch.zeros_like(out)
            ys = range(0, h - overlap, tile - overlap) if h > tile else [0]
            xs = range(0, w - overlap, tile - overlap) if w > tile else [0]
            for y in ys:
                for x in xs:
                    py, px = max(0, min(h - overlap, y)), max(0, min(w - overlap, x))
                    ly, lx = min(tile, h - py), min(tile, w - px)
                    ps = self.forward(img[:, py:py + ly, px:px + lx, :].contiguous())
                    lib.check(L.ldx_tile_blend(lib.ptr(ps), s * ly, s * lx, lib.ptr(out), lib.ptr(div), s * h, s * w, self.cfg.out_nc,
                                               round(s * py), round(s * px), round(s * overlap), lib.current_stream_ptr()), "ldx_tile_blend")
            lib.check(L.ldx_tile_finish(lib.ptr(out), lib.ptr(div), out.numel(), 1, lib.current_stream_ptr()), "ldx_tile_finish")
            outs.append(out)
        return torch.cat(outs, 0)

    profile = UNetEngine.profile
    profile_report = UNetEngine.profile_report
    plan_info = UNetEngine.plan_info
    close = UNetEngine.close
    __del__ = UNetEngine.__del__


def t5_relative_position_bucket(relative_position, num_buckets=32, max_distance=128):
    """T5Attention._relative_position_bucket, bidirectional (src/clip/FluxClip.py:152-205) — same torch ops in the same
    order, so bucket boundaries (float32 log) agree bit for bit with the reference."""
    import math as _m
    num_buckets //= 2
    relative_buckets = (relative_position > 0).to(torch.long) * num_buckets
    relative_position = torch.abs(relative_position)
    max_exact = num_buckets // 2
    is_small = relative_position < max_exact
    large = max_exact + (torch.log(relative_position.float() / max_exact) / _m.log(max_distance / max_exact)
                         * (num_buckets - max_exact)).to(torch.long)
    large = torch.min(large, torch.full_like(large, num_buckets - 1))
    return relative_buckets + torch.where(is_small, relative_position, large)


def t5_bias_table(rel_bias_weight, length: int, num_buckets=32, max_distance=128):
    """T5Attention.compute_bias (FluxClip.py:207-243) -> fp32 [H][L][Lp], Lp = L rounded up to 64 (zero padded): the
    layout ldx_t5_encode reads."""
    ctx = torch.arange(length, dtype=torch.long)[:, None]
    mem = torch.arange(length, dtype=torch.long)[None, :]
    bucket = t5_relative_position_bucket(mem - ctx, num_buckets, max_distance)           # [L, L]
    vals = rel_bias_weight.float()[bucket].permute(2, 0, 1).contiguous()                 # [H, L, L]
    lp = (length + 63) // 64 * 64
    out = torch.zeros((vals.shape[0], length, lp), dtype=torch.float32)
    out[:, :, :length] = vals
    return out


class T5Engine:
    """T5.forward (src/clip/FluxClip.py:441-519): token ids -> final_layer_norm(encoder output), fp32.  Host side keeps
    the 32 x heads relative-attention embedding and builds the bias table per sequence length (cached)."""

    def __init__(self, cfg, state_dict, device: int = 0, dtype: str = "bf16"):
        self._lib = lib.load()
        self._h = C.c_void_p()
        self.cfg, self.device = cfg, torch.device("cuda", device)
        c = lib.ldx_t5_config()
        c.compute_dtype = {"bf16": lib.LDX_BF16, "f16": lib.LDX_F16, "fp16": lib.LDX_F16}[dtype]
        c.d_model, c.d_ff, c.num_layers, c.num_heads, c.vocab_size = cfg.d_model, cfg.d_ff, cfg.num_layers, cfg.num_heads, cfg.vocab_size
        lib.check(self._lib.ldx_t5_create(C.byref(c), device, C.byref(self._h)), "ldx_t5_create")
        rk = "encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"
        self._rel = state_dict[rk].float().cpu()
        _load_state_dict(self._lib, self._h, {k: v for k, v in state_dict.items() if k != rk}, strip=("t5xxl.transformer.",))
        lib.check(self._lib.ldx_finalize(self._h), "ldx_finalize")
        self._bias = {}

    def bias(self, length: int):
        if length not in self._bias:
            self._bias[length] = t5_bias_table(self._rel, length, self.cfg.num_buckets, self.cfg.max_distance).to(self.device)
        return self._bias[length]

    def forward(self, tokens):
        """tokens: int tensor [B][L] -> [B, L, d_model] fp32."""
        ids = tokens.to(self.device, torch.int32).contiguous()
        b, l = ids.shape
        out = torch.empty((b, l, self.cfg.d_model), device=self.device, dtype=torch.float32)
        lib.check(self._lib.ldx_t5_encode(self._h, lib.ptr(ids), b, l, lib.ptr(self.bias(l)), lib.ptr(out), lib.current_stream_ptr()),
                  "ldx_t5_encode")
        return out

    def encode_token_weights(self, token_weight_pairs):
        """ClipTokenWeightEncoder.encode_token_weights (SDClip.py:36-97) as T5XXLModel runs it (FluxClip.py:521-545):
        special tokens {end: 1, pad: 0}, layer "last", no pooled output.  Returns (cond [1, sum L, d_model], None)."""
        to_encode, has_weights, max_len = [], False, 0
        for x in token_weight_pairs:
            toks = [a[0] for a in x]
            max_len = max(max_len, len(toks))
            has_weights = has_weights or not all(a[1] == 1.0 for a in x)
            to_encode.append(toks)
        sections = len(to_encode)
        if has_weights or sections == 0:
            to_encode.append([1] + [0] * (max_len - 1))                      # gen_empty_tokens (SDClip.py:10-22)
        out = self.forward(torch.tensor(to_encode, dtype=torch.int64)).float().cpu()
        output = []
        for k in range(sections):
            z = out[k:k + 1].clone()
            if has_weights:
                z_empty = out[-1]
                for j in range(z.shape[1]):
                    wgt = token_weight_pairs[k][j][1]
                    if wgt != 1.0:
                        z[0][j] = (z[0][j] - z_empty[j]) * wgt + z_empty[j]
            output.append(z)
        cond = out[-1:] if not output else torch.cat(output, dim=-2)
        return cond, None

    profile = UNetEngine.profile
    profile_report = UNetEngine.profile_report
    plan_info = UNetEngine.plan_info
    close = UNetEngine.close
    __del__ = UNetEngine.__del__


def flux_rope_tables(cfg: FluxConfig, batch_txt_len: int, h: int, w: int):
    """pe for ids = [txt_ids (zeros) ; img_ids] exactly as Flux3.forward + EmbedND + rope() build it
    (src/BlackForest/Flux.py:36-70, 96-103, 750-771): returns (cos, sin) fp32 [Lt + (h/2)(w/2)][head_dim/2]."""
    h_len, w_len = (h + 1) // 2, (w + 1) // 2
    img_ids = torch.zeros((h_len, w_len, 3), dtype=torch.float32)
    img_ids[..., 1] = img_ids[..., 1] + torch.linspace(0, h_len - 1, steps=h_len, dtype=torch.float32)[:, None]
    img_ids[..., 2] = img_ids[..., 2] + torch.linspace(0, w_len - 1, steps=w_len, dtype=torch.float32)[None, :]
    ids = torch.cat([torch.zeros((batch_txt_len, 3), dtype=torch.float32), img_ids.reshape(-1, 3)], dim=0)
    cos, sin = [], []
    for i, dim in enumerate(cfg.axes_dim):
        scale = torch.linspace(0, (dim - 2) / dim, steps=dim // 2, dtype=torch.float64)
        omega = 1.0 / (cfg.theta ** scale)
        out = torch.einsum("n,d->nd", ids[:, i].to(torch.float32), omega)          # float32 x float64 -> float64, as rope()
        cos.append(torch.cos(out).to(torch.float32)); sin.append(torch.sin(out).to(torch.float32))
    return torch.cat(cos, dim=-1).contiguous(), torch.cat(sin, dim=-1).contiguous()


class FluxEngine:
    """Flux3.forward behind BaseModel.apply_model with CONST prediction."""

    def __init__(self, cfg: FluxConfig, state_dict, device: int = 0, dtype: str = "bf16", fp8: bool = False):
        """fp8=True (or "linears"): the block linears (and the adaLN modulation projections) run on MX fp8 operands, attention in 16 bit (ldx_flux_set_fp8 mode 1;
        approximate, opt-in, own parity class).  fp8="attn": explicit opt-in to the less precise full mode (mode 3): at head dim 128 QK^T / PV of the joint attention run on MX fp8 too."""
        self._lib = lib.load()
        self._h = C.c_void_p()
        self.cfg, self.device = cfg, torch.device("cuda", device)
        c = lib.ldx_flux_config()
        c.compute_dtype = {"bf16": lib.LDX_BF16, "f16": lib.LDX_F16, "fp16": lib.LDX_F16}[dtype]
        c.in_channels, c.vec_in_dim, c.context_in_dim = cfg.in_channels, cfg.vec_in_dim, cfg.context_in_dim
        c.hidden_size, c.mlp_hidden, c.num_heads = cfg.hidden_size, cfg.mlp_hidden, cfg.num_heads
        c.depth, c.depth_single, c.guidance_embed = cfg.depth, cfg.depth_single_blocks, int(cfg.guidance_embed)
        lib.check(self._lib.ldx_flux_create(C.byref(c), device, C.byref(self._h)), "ldx_flux_create")
        if fp8:
            assert fp8 in (True, "linears", "attn"), "fp8 must be False, True / 'linears', or 'attn'"
            lib.check(self._lib.ldx_flux_set_fp8(self._h, 3 if fp8 == "attn" else 1), "ldx_flux_set_fp8")
        _load_state_dict(self._lib, self._h, state_dict, strip=("model.diffusion_model.",))
        lib.check(self._lib.ldx_finalize(self._h), "ldx_finalize")
        self._pe = {}

    def _run(self, x, sigma, ctx, y, guidance, denoise):
        assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 4
        b, _, h, w = x.shape
        lt = ctx.shape[1]
        key = (lt, h, w)
        if key not in self._pe:
            cos, sin = flux_rope_tables(self.cfg, lt, h, w)
            self._pe[key] = (cos.to(self.device), sin.to(self.device))
        cos, sin = self._pe[key]
        f32 = lambda t: None if t is None else t.to(self.device, torch.float32).contiguous()
        x, sigma, ctx, y, guidance = f32(x), f32(sigma), f32(ctx), f32(y), f32(guidance)
        out = torch.empty_like(x)
        lib.check(self._lib.ldx_flux_forward(self._h, lib.ptr(x), lib.ptr(sigma), lib.ptr(ctx), lib.ptr(y), lib.ptr(guidance),
                                             lib.ptr(cos), lib.ptr(sin), b, h, w, lt, int(denoise), lib.ptr(out),
                                             lib.current_stream_ptr()), "ldx_flux_forward")
        return out

    def set_fbcache(self, residual_diff_threshold: float):
        """ApplyFBCacheOnModel.patch (fbcache_nodes.py:8-201): opt-in approximate first-block cache; 0 disables."""
        lib.check(self._lib.ldx_flux_fbcache(self._h, float(residual_diff_threshold)), "ldx_flux_fbcache")

    def fbcache_stats(self):
        h, m = C.c_int64(), C.c_int64()
        lib.check(self._lib.ldx_flux_fbcache_stats(self._h, C.byref(h), C.byref(m)), "ldx_flux_fbcache_stats")
        return {"hits": h.value, "misses": m.value}

    def forward(self, x, timestep, ctx, y, guidance):
        """Flux3.forward(x, timestep, context, y, guidance) (Flux.py:732-778)."""
        return self._run(x, timestep, ctx, y, guidance, False)

    def denoise(self, x, sigma, ctx, y, guidance):
        """BaseModel.apply_model for ModelType.FLUX: x - Flux3(x, sigma, ...) * sigma (sampling.py:100-155)."""
        return self._run(x, sigma, ctx, y, guidance, True)

    profile = UNetEngine.profile
    profile_report = UNetEngine.profile_report
    plan_info = UNetEngine.plan_info
    close = UNetEngine.close
    __del__ = UNetEngine.__del__
