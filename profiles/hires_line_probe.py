"""bench.py's HiresFix secondary line alone (A/B of planner / sampler switches through the environment)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import ldx_amd as ldx
cfg = ldx.UNetConfig.sd15()
unet = ldx.UNetEngine(cfg, ldx.weights.synth_state_dict(ldx.weights.unet_state_dict_spec(cfg), seed=1234), dtype="bf16")
unet.set_graph_mode(True)
r = bench.hiresfix_line(ldx, unet, cfg)
print({k: r[k] for k in ("sampler_ms", "unet_evaluations", "ms_per_evaluation", "vae_decode_2048_ms", "esrgan_tile_ms")}, unet.graph_stats())
