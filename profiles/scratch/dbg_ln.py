import os, sys, torch
sys.path.insert(0, "/root/repo")
import ldx_amd as ldx
cfg = ldx.UNetConfig.sd15()
sd = ldx.weights.synth_state_dict(ldx.weights.unet_state_dict_spec(cfg), seed=1234)
eng = ldx.UNetEngine(cfg, sd, dtype="bf16")
for lat in (128,):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 4, lat, lat, generator=g).cuda(); sig = torch.tensor([5.0, 5.0]).cuda(); ctx = torch.randn(2, 77, 768, generator=g).cuda()
    out = eng.denoise(x, sig, ctx)
    print(lat, "nan frac per batch", [float(torch.isnan(out[b]).float().mean()) for b in range(2)], flush=True)
