"""ldx — MI355X-native engine for LightDiffusion-Next's denoising hot path.

The package directory is named ``lightdiffusion-next_amd`` (not a Python identifier); import it as
``import ldx_amd`` (shim at the repository root) or ``importlib.import_module("lightdiffusion-next_amd")``.

Layout:
  csrc/        hand-written HIP kernels for gfx950 + the C ABI (include/ldx.h) -> libldx.so
  lib.py       ctypes binding of libldx.so (fails loudly if the library or a GPU is missing)
  engine.py    UNetEngine: weights in, denoise out (device pointers through the C ABI)
  hook.py      LdxUNetPatch / LdxFluxPatch: drop-ins for model_options["model_function_wrapper"] (cond.py:254-263)
  sampling.py  host mirror of src/sample (schedulers, CFG batching, Euler / DPM++ loops)
  parallel.py  batch shard + single all-gather across the GPUs of a node (RCCL / gloo)
  weights.py   SD1.5 state-dict layout + seeded synthetic weights (no checkpoints offline)
  checkpoint.py  load-time ingestion: checkpoint split, UNet layout sniffing, LoRA key maps + merge
"""
from . import lib, weights  # noqa: F401
from .engine import UNetEngine, UNetConfig, VAEDecoderEngine, CLIPTextEngine, FluxEngine, T5Engine, ESRGANEngine, bislerp, latent_upscale  # noqa: F401
VAEEngine = VAEDecoderEngine      # the same engine encodes when encoder.* weights are loaded
from .weights import VAEConfig, CLIPConfig, FluxConfig, T5Config, ESRGANConfig  # noqa: F401
from .hook import LdxUNetPatch, LdxFluxPatch  # noqa: F401
from . import prompt  # noqa: F401
from . import sampling, parallel, checkpoint  # noqa: F401
