"""Per-op-shape timing of one full-size SD1.5 forward (1024^2, CFG batch 2): HIP events per op (ldx_profile mode 2).
Usage: python profiles/shape_probe.py [latent=128] [dtype=bf16]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ldx_amd as ldx

lat = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dt = sys.argv[2] if len(sys.argv) > 2 else "bf16"
cfg = ldx.UNetConfig.sd15()
sd = ldx.weights.synth_state_dict(ldx.weights.unet_state_dict_spec(cfg), seed=1234)
eng = ldx.UNetEngine(cfg, sd, dtype=dt)
x = torch.randn(2, 4, lat, lat, device="cuda"); sig = torch.full((2,), 5.0, device="cuda"); ctx = torch.randn(2, 77, 768, device="cuda")
out = torch.empty_like(x)
for _ in range(3):
    eng.denoise(x, sig, ctx, out=out)
torch.cuda.synchronize()
eng._lib.ldx_profile(eng._h, 2, 1)
n = 5
for _ in range(n):
    eng.denoise(x, sig, ctx, out=out)
torch.cuda.synchronize()
eng._lib.ldx_profile(eng._h, 0, 0)
rep = eng.profile_report()
tot = sum(v["ms"] for v in rep.values()) / n
print(f"sum of op times {tot:.3f} ms / forward")
for k, v in sorted(rep.items(), key=lambda kv: -kv[1]["ms"]):
    c = v["count"] // n
    ms = v["ms"] / n
    tf = f"{v['flops'] / v['ms'] / 1e9:7.0f} TF" if v["flops"] else "          "
    gb = f"{v['bytes'] / v['ms'] / 1e6:6.0f} GB/s" if v["bytes"] else ""
    print(f"{ms:8.3f} ms {100 * ms / tot:5.1f}%  n={c:3d} {1e3 * ms / max(c, 1):8.1f} us/op {tf} {gb}  {k}")
