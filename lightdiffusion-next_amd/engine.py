"""UNetEngine — Python handle on the native SD1.5 UNet engine (libldx.so).

Host-side responsibilities only: hand the state dict to the C ABI once, build the two lookup tables the
reference builds on the host (sigma table and sinusoidal timestep embeddings), and pass device pointers.
Reference counterparts: BaseModel.__init__/load_model_weights/apply_model (src/Model/ModelBase.py:38-202).
"""
import ctypes as C
import math

import torch

from . import lib
from .weights import UNetConfig


def sd15_sigmas():
    """ModelSamplingDiscrete._register_schedule + set_sigmas (src/sample/sampling.py:224-289) with
    make_beta_schedule("linear", 1000, 0.00085, 0.012) (src/sample/sampling_util.py:18-39).
    Returns (sigmas fp32 [1000], log_sigmas fp32 [1000]); fp64 until the final cast, like the reference."""
    betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=torch.float64) ** 2
    alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
    sigmas = ((1 - alphas_cumprod) / alphas_cumprod) ** 0.5
    return sigmas.float(), sigmas.log().float()


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

    def _run(self, fn, x, s, ctx, out):
        assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 4, "x must be a CUDA fp32 NCHW tensor"
        b2, ch, h, w = x.shape
        assert ch == self.cfg.in_channels
        x = x.contiguous()
        s = s.to(device=x.device, dtype=torch.float32).contiguous()
        ctx = ctx.to(device=x.device, dtype=torch.float32).contiguous()
        assert s.numel() == b2 and ctx.dim() == 3 and ctx.shape[0] == b2 and ctx.shape[2] == self.cfg.context_dim
        if out is None:
            out = torch.empty((b2, self.cfg.out_channels, h, w), device=x.device, dtype=torch.float32)
        lib.check(fn(self._h, lib.ptr(x), lib.ptr(s), lib.ptr(ctx), b2, h, w, ctx.shape[1], lib.ptr(out),
                     lib.current_stream_ptr()), fn.__name__)
        return out

    def denoise(self, x, sigma, ctx, out=None):
        """BaseModel.apply_model (ModelBase.py:72-133): x fp32 [B2,4,h,w], sigma [B2] (values), ctx [B2,M,768]."""
        return self._run(self._lib.ldx_unet_denoise, x, sigma, ctx, out)

    def forward(self, x, timesteps, ctx, out=None):
        """UNetModel1.forward (unet.py:679-770): integer timesteps (as floats), unscaled input."""
        return self._run(self._lib.ldx_unet_forward, x, timesteps, ctx, out)

    def profile(self, on: bool, reset: bool = True):
        lib.check(self._lib.ldx_profile(self._h, int(on), int(reset)), "ldx_profile")

    def profile_report(self) -> dict:
        import json
        buf = C.create_string_buffer(1 << 16)
        lib.check(self._lib.ldx_profile_report(self._h, buf, len(buf)), "ldx_profile_report")
        return json.loads(buf.value.decode())

    def plan_info(self):
        n, f, a = C.c_int64(), C.c_double(), C.c_int64()
        lib.check(self._lib.ldx_plan_info(self._h, C.byref(n), C.byref(f), C.byref(a)), "ldx_plan_info")
        return {"launches": n.value, "flops": f.value, "arena_bytes": a.value}
