"""End-to-end latency of the reference pipeline's SD1.5 shapes on the engines (synthetic weights): CLIP encode of both
prompts, KSampler with the pipeline's default sampler (dpmpp_sde_cfgpp / karras, 20 steps, cfg 7 = 39 CFG-batched UNet
evaluations, pipeline.py:114,326) and with the reference-default "euler" name (forced multi-scale: 9 of 20 steps at half
resolution), VAE decode.  Usage: python profiles/pipeline_probe.py [latent=128]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ldx_amd as ldx

lat = int(sys.argv[1]) if len(sys.argv) > 1 else 128
ucfg = ldx.UNetConfig.sd15()
unet = ldx.UNetEngine(ucfg, ldx.weights.synth_state_dict(ldx.weights.unet_state_dict_spec(ucfg), seed=1234), dtype="bf16")
vcfg = ldx.VAEConfig()
vae = ldx.VAEDecoderEngine(vcfg, ldx.weights.synth_state_dict(ldx.weights.vae_decoder_state_dict_spec(vcfg), seed=1, dtype=torch.float32), dtype="bf16")
ccfg = ldx.CLIPConfig()
clip = ldx.CLIPTextEngine(ccfg, ldx.weights.synth_state_dict(ldx.weights.clip_state_dict_spec(ccfg), seed=2), dtype="bf16")
unet.set_graph_mode(True)
ks = ldx.sampling.KSampler(unet)
ids = torch.randint(0, 49407, (2, 77))


def run(sampler, scheduler, **kw):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    cond = clip.forward(ids, -2)
    cond = cond[0] if isinstance(cond, (tuple, list)) else cond
    pos, neg = cond[0:1].float(), cond[1:2].float()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    trace = []
    x = ks.sample(seed=1, steps=20, cfg=7.0, sampler_name=sampler, scheduler=scheduler, positive=pos, negative=neg,
                  latent_image=torch.zeros(1, 4, lat, lat), trace=trace, **kw)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    img = vae.decode(x)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    return t1 - t0, t2 - t1, t3 - t2, len(trace), bool(torch.isfinite(img).all())


for name, sched, kw in (("dpmpp_sde_cfgpp", "karras", {}), ("euler", "normal", {}), ("sample_euler", "normal", dict(enable_multiscale=False)),
                        ("dpmpp_2m_cfgpp", "karras", {})):
    run(name, sched, **kw)
    c, s, v, n, ok = run(name, sched, **kw)
    print(f"{name:18s}/{sched:6s}: CLIP {c * 1e3:6.2f} ms | sampler {s * 1e3:7.1f} ms ({n} UNet evaluations, {n / s:5.1f} eval/s) | VAE decode {v * 1e3:5.1f} ms | "
          f"image {8 * lat}^2 in {(c + s + v):.3f} s  finite={ok}", flush=True)

# ---- BASELINE config 5 shape: txt2img latents -> bislerp x2 -> 10 steps euler_ancestral_cfgpp / normal, denoise 0.45 at 2048^2
#      (pipeline.py:346-366) -> VAE decode 2048^2 (untiled) ----
if lat == 128 and os.environ.get("LDX_SKIP_HIRES", "0") != "1":
    cond = clip.forward(ids, -2)
    cond = cond[0] if isinstance(cond, (tuple, list)) else cond
    pos, neg = cond[0:1].float(), cond[1:2].float()
    base = ks.sample(seed=1, steps=20, cfg=7.0, sampler_name="dpmpp_sde_cfgpp", scheduler="karras", positive=pos, negative=neg,
                     latent_image=torch.zeros(1, 4, lat, lat))
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        up = ldx.latent_upscale(base, 2 * 8 * lat, 2 * 8 * lat)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        hi = ks.sample(seed=2, steps=10, cfg=8.0, denoise=0.45, sampler_name="euler_ancestral_cfgpp", scheduler="normal", positive=pos, negative=neg,
                       latent_image=up)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        img = vae.decode(hi)
        torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f"HiresFix 2048^2: bislerp {1e3 * (t1 - t0):.1f} ms | 10 steps euler_ancestral_cfgpp (denoise 0.45 -> {len(ldx.sampling.sigmas_for(ks.model_sampling, 'normal', 10, 0.45)) - 1} evaluations) "
          f"{1e3 * (t2 - t1):.0f} ms | VAE decode {1e3 * (t3 - t2):.0f} ms | total {(t3 - t0):.2f} s  finite={bool(torch.isfinite(img).all())}")
