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
