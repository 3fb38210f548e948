import sys, time, torch
sys.path.insert(0, "/root/repo")
import ldx_amd as ldx
cfg = ldx.UNetConfig.sd15()
sd = ldx.weights.synth_state_dict(ldx.weights.unet_state_dict_spec(cfg), seed=1234)
for graph in (False, True):
    eng = ldx.UNetEngine(cfg, sd, dtype="bf16", graph=graph)
    x = torch.randn(2, 4, 256, 256, device="cuda"); sig = torch.full((2,), 5.0, device="cuda"); ctx = torch.randn(2, 77, 768, device="cuda"); out = torch.empty_like(x)
    for _ in range(3): eng.denoise(x, sig, ctx, out=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): eng.denoise(x, sig, ctx, out=out)
    torch.cuda.synchronize(); print("graph", graph, (time.perf_counter() - t0) * 100, "ms per evaluation", flush=True)
    del eng
ks_eng = ldx.UNetEngine(cfg, sd, dtype="bf16", graph=True)
ks = ldx.sampling.KSampler(ks_eng)
pos = torch.randn(1, 77, 768); neg = torch.randn(1, 77, 768)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    hi = ks.sample(seed=2, steps=10, cfg=8.0, denoise=0.45, sampler_name="euler_ancestral_cfgpp", scheduler="normal", positive=pos, negative=neg, latent_image=torch.randn(1, 4, 256, 256))
    torch.cuda.synchronize(); print("euler_ancestral_cfgpp 10 steps denoise .45:", time.perf_counter() - t0, "s", flush=True)
