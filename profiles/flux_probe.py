"""Flux-dev DiT forward at 1024x1024 (latent 128x128 -> 4096 img tokens + 256 txt tokens), synthetic weights.
BASELINE config 4 shape in bf16 (fp8 not built).  Prints ms / forward, TFLOP/s and the per-kernel profile."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ldx_amd as ldx

depth = int(sys.argv[1]) if len(sys.argv) > 1 else 19
single = int(sys.argv[2]) if len(sys.argv) > 2 else 38
cfg = ldx.FluxConfig(depth=depth, depth_single_blocks=single)
t0 = time.time()
spec = ldx.weights.flux_state_dict_spec(cfg)
sd = {}
g = torch.Generator().manual_seed(1)
for k, shp in spec:        # cheap fill: one small random block tiled (values do not matter for timing; still random data)
    n = 1
    for d in shp: n *= d
    if k.endswith(".bias"): sd[k] = (0.02 * torch.randn(shp, generator=g)).half()
    elif k.endswith("scale"): sd[k] = torch.ones(shp).half()
    else:
        base = torch.randn(min(n, 1 << 20), generator=g) / (shp[-1] ** 0.5)
        sd[k] = base.repeat((n + base.numel() - 1) // base.numel())[:n].reshape(shp).half()
print(f"weights: {ldx.weights.param_count(spec)/1e9:.2f} B params generated in {time.time()-t0:.1f} s", flush=True)
t0 = time.time()
FP8 = {"0": False, "1": "attn", "linears": True}[os.environ.get("LDX_FLUX_FP8", "0")]      # 1: the full mode (linears + attention, bench config 4); linears: attention in 16 bit
eng = ldx.FluxEngine(cfg, sd, dtype="bf16", fp8=FP8)
print("fp8 (MX) mode:", FP8)
del sd
print(f"engine built in {time.time()-t0:.1f} s", flush=True)
x = torch.randn(1, 16, 128, 128, device="cuda"); ctx = torch.randn(1, 256, 4096, device="cuda"); y = torch.randn(1, 768, device="cuda")
t = torch.tensor([0.7], device="cuda"); gd = torch.tensor([3.0], device="cuda")
for _ in range(2): out = eng.denoise(x, t, ctx, y, gd)
torch.cuda.synchronize()
n = 5
t0 = time.perf_counter()
for _ in range(n): out = eng.denoise(x, t, ctx, y, gd)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
info = eng.plan_info()
print(f"Flux DiT forward 1024^2 bs1 depth {depth}+{single}: {dt*1e3:.1f} ms  {info['flops']/dt/1e12:.0f} TFLOP/s ({info['flops']/1e12:.1f} TFLOP)  "
      f"launches {info['launches']} arena {info['arena_bytes']/2**30:.2f} GiB finite={bool(torch.isfinite(out).all())}")
if os.environ.get("LDX_PROBE_SHAPES") == "1":      # per-op-shape table (ldx_profile mode 2) instead of the per-class one
    eng._lib.ldx_profile(eng._h, 2, 1); eng.denoise(x, t, ctx, y, gd); torch.cuda.synchronize(); eng._lib.ldx_profile(eng._h, 0, 0)
    for k, v in sorted(eng.profile_report().items(), key=lambda kv: -kv[1]["ms"])[:30]:
        print(f"  {v['ms']:8.3f} ms  n={v['count']:4d} {1e3 * v['ms'] / max(v['count'], 1):8.1f} us/op" + (f"  {v['flops']/v['ms']/1e9:6.0f} TF" if v['flops'] else "          ") + f"  {k}")
    sys.exit(0)
eng.profile(True); eng.denoise(x, t, ctx, y, gd); torch.cuda.synchronize(); eng.profile(False, reset=False)
for k, v in sorted(eng.profile_report().items(), key=lambda kv: -kv[1]["ms"])[:8]:
    print(f"  {k:30s} n={v['count']:4d} {v['ms']:.2f} ms" + (f"  {v['flops']/v['ms']/1e9:.0f} TF" if v['flops'] else ""))

# ---- the reference's Flux pipeline shape (pipeline.py:237-262): 20 steps euler_cfgpp / beta, cfg 1 with a zeroed negative
# (both branches evaluated: batch 2), guidance 3.0, optional first-block cache 0.12 ----
if len(sys.argv) > 3 and sys.argv[3] == "sampler":
    ks = ldx.sampling.FluxKSampler(eng)
    pos = (torch.randn(1, 256, 4096), torch.randn(1, 768))
    neg = (torch.zeros(1, 256, 4096), torch.zeros(1, 768))
    for thr in (0.0, 0.12):
        eng.set_fbcache(thr)
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            out = ks.sample(seed=1, steps=20, cfg=1, sampler_name="euler_cfgpp", scheduler="beta", positive=pos, negative=neg,
                            latent_image=torch.zeros(1, 16, 128, 128), guidance=3.0)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        st = eng.fbcache_stats()
        print(f"Flux KSampler 20 steps euler_cfgpp/beta 1024^2 (22 batch-2 forwards), fbcache {thr}: {dt:.2f} s  "
              f"({20 / dt:.2f} it/s)  cache hits {st['hits']}/{st['hits'] + st['misses']} finite={bool(torch.isfinite(out).all())}")
