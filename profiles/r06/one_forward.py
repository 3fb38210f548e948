"""A few EAGER CFG evaluations of the headline workload (SD1.5 1024^2, latent [1,4,128,128], CFG batch 2, the sampler loops' entry point ldx_unet_denoise_cfg_t:
the plan with the shared CFG prefix) and nothing else — the process rocprofv3 --pmc / --kernel-trace passes run (profiles/pmc_r06.sh): every kernel is its own
dispatch, in plan order; each forward starts with prep_image_kernel.  Usage: python profiles/r06/one_forward.py [forwards=3] [latent=128] [pb=1]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ldx_amd as ldx

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
lat = int(sys.argv[2]) if len(sys.argv) > 2 else 128
pb = int(sys.argv[3]) if len(sys.argv) > 3 else 1
cfg = ldx.UNetConfig.sd15()
sd = ldx.weights.synth_state_dict(ldx.weights.unet_state_dict_spec(cfg), seed=1234)
eng = ldx.UNetEngine(cfg, sd, dtype="bf16")
del sd
g = torch.Generator().manual_seed(7)
x = torch.randn(pb, 4, lat, lat, generator=g).cuda(); ctx = torch.randn(2 * pb, 77, 768, generator=g).cuda()
out = torch.empty(2 * pb, 4, lat, lat, device="cuda")
eng.set_context_cache(True)                 # steady state of a sampling run: the context's k|v projections are computed once
for _ in range(n):
    eng.denoise_cfg(x, 5.0, ctx, out=out, ctx_cached=True)
torch.cuda.synchronize()
info = eng.plan_info()
print(f"one_forward: {n} evaluations, {info['launches']} launches each, {info['flops_executed'] / 1e12:.3f} of {info['flops'] / 1e12:.3f} TFLOP executed, finite={bool(torch.isfinite(out).all())}")
