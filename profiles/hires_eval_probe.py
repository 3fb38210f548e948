"""Per-evaluation GPU time inside the HiresFix sampler loop (euler_ancestral_cfgpp at latent 256^2), by HIP events around UNetEngine.denoise_cfg."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ldx_amd as ldx
cfg = ldx.UNetConfig.sd15()
unet = ldx.UNetEngine(cfg, ldx.weights.synth_state_dict(ldx.weights.unet_state_dict_spec(cfg), seed=1234), dtype="bf16")
unet.set_graph_mode(os.environ.get("GRAPH", "1") == "1")
evs = []
orig = unet.denoise_cfg
def timed(*a, **k):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = orig(*a, **k); e1.record(); evs.append((e0, e1)); return r
unet.denoise_cfg = timed
ks = ldx.sampling.KSampler(unet)
g = torch.Generator().manual_seed(5)
pos, neg = torch.randn([1, 77, 768], generator=g), torch.randn([1, 77, 768], generator=g)
up = torch.randn(1, 4, 256, 256, device="cuda")
for rep in range(3):
    evs.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    hi = ks.sample(seed=2, steps=10, cfg=8.0, denoise=0.45, sampler_name="euler_ancestral_cfgpp", scheduler="normal", positive=pos, negative=neg, latent_image=up)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"rep {rep}: sampler {1e3 * (t1 - t0):.1f} ms; per evaluation (GPU events): " + " ".join(f"{a.elapsed_time(b):.1f}" for a, b in evs), unet.graph_stats())
