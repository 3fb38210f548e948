"""Reference-default "euler" (forced multi-scale: 9 of 20 steps at half resolution) through KSampler: it/s and hipGraph capture / replay counts."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ldx_amd as ldx
cfg = ldx.UNetConfig.sd15()
eng = ldx.UNetEngine(cfg, ldx.weights.synth_state_dict(ldx.weights.unet_state_dict_spec(cfg), seed=1234), dtype="bf16")
eng.set_graph_mode(True)
ks = ldx.sampling.KSampler(eng)
pos = torch.randn(1, 77, 768); neg = torch.randn(1, 77, 768)
for rep in range(3):
    c0 = eng.graph_stats()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    x = ks.sample(seed=1, steps=20, cfg=7.0, sampler_name="euler", scheduler="normal", positive=pos, negative=neg, latent_image=torch.zeros(1, 4, 128, 128))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    c1 = eng.graph_stats()
    print(f"run {rep}: {dt * 1e3:.1f} ms  {20 / dt:.2f} it/s  graph captures +{c1[0] - c0[0]} replays +{c1[1] - c0[1]}")
