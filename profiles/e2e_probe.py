"""Time VAE decode (1024^2) and CLIP-L encode through the C ABI; print per-kernel profile for the VAE."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ldx_amd as ldx

lat = int(sys.argv[1]) if len(sys.argv) > 1 else 128
vcfg = ldx.VAEConfig()
vsd = ldx.weights.synth_state_dict(ldx.weights.vae_decoder_state_dict_spec(vcfg), seed=1, dtype=torch.float32)
vae = ldx.VAEDecoderEngine(vcfg, vsd, dtype="bf16")
z = torch.randn(1, 4, lat, lat, device="cuda")
for _ in range(2):
    img = vae.decode(z)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 5
for _ in range(n):
    img = vae.decode(z)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
info = vae.plan_info()
print(f"VAE decode latent {lat}x{lat}: {dt*1e3:.2f} ms  {info['flops']/dt/1e12:.1f} TFLOP/s  launches {info['launches']} arena {info['arena_bytes']/2**30:.2f} GiB  finite={bool(torch.isfinite(img).all())}")
vae.profile(True)
vae.decode(z); torch.cuda.synchronize()
vae.profile(False, reset=False)
rep = vae.profile_report()
for k, v in sorted(rep.items(), key=lambda kv: -kv[1]["ms"])[:8]:
    print(f"  {k:28s} n={v['count']:3d} {v['ms']:.3f} ms" + (f"  {v['flops']/v['ms']/1e9:.0f} TF" if v['flops'] else ""))
ccfg = ldx.CLIPConfig()
csd = ldx.weights.synth_state_dict(ldx.weights.clip_state_dict_spec(ccfg), seed=2)
clip = ldx.CLIPTextEngine(ccfg, csd, dtype="bf16")
ids = torch.randint(0, 49407, (2, 77))
for _ in range(2):
    clip.forward(ids, -2)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    clip.forward(ids, -2)
torch.cuda.synchronize()
print(f"CLIP-L encode 2x77 tokens: {(time.perf_counter()-t0)/10*1e3:.2f} ms")
