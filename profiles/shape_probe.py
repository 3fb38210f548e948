"""Per-op-shape timing of one full-size SD1.5 forward (1024^2, CFG batch 2 by default): HIP events per op (ldx_profile mode 2).
Usage: python profiles/shape_probe.py [latent=128] [dtype=bf16] [cfg_batch=2] [cfg]      (cfg_batch 16 = BASELINE config 3's per-GPU shard;
"cfg": through ldx_unet_denoise_cfg_t like the sampler loops — the plan with the shared CFG prefix — instead of ldx_unet_denoise on a full batch)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ldx_amd as ldx

lat = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dt = sys.argv[2] if len(sys.argv) > 2 else "bf16"
B2 = int(sys.argv[3]) if len(sys.argv) > 3 else 2
CFG = len(sys.argv) > 4 and sys.argv[4] == "cfg"
cfg = ldx.UNetConfig.sd15()
sd = ldx.weights.synth_state_dict(ldx.weights.unet_state_dict_spec(cfg), seed=1234)
eng = ldx.UNetEngine(cfg, sd, dtype=dt)
x = torch.randn(B2, 4, lat, lat, device="cuda"); sig = torch.full((B2,), 5.0, device="cuda"); ctx = torch.randn(B2, 77, 768, device="cuda")
out = torch.empty_like(x)
xh = x[:B2 // 2].contiguous()
run = (lambda: eng.denoise_cfg(xh, 5.0, ctx, out=out)) if CFG else (lambda: eng.denoise(x, sig, ctx, out=out))
for _ in range(3):
    run()
torch.cuda.synchronize()
eng._lib.ldx_profile(eng._h, 2, 1)
n = 5
for _ in range(n):
    run()
torch.cuda.synchronize()
eng._lib.ldx_profile(eng._h, 0, 0)
rep = eng.profile_report()
tot = sum(v["ms"] for v in rep.values()) / n
info = eng.plan_info()
print(f"sum of op times {tot:.3f} ms / forward   ({'denoise_cfg' if CFG else 'denoise'}, {info['launches']} launches, {info['flops_executed'] / 1e12:.3f} of {info['flops'] / 1e12:.3f} TFLOP executed)")
for k, v in sorted(rep.items(), key=lambda kv: -kv[1]["ms"]):
    c = v["count"] // n
    ms = v["ms"] / n
    tf = f"{v['flops'] / v['ms'] / 1e9:7.0f} TF" if v["flops"] else "          "
    gb = f"{v['bytes'] / v['ms'] / 1e6:6.0f} GB/s" if v["bytes"] else ""
    print(f"{ms:8.3f} ms {100 * ms / tot:5.1f}%  n={c:3d} {1e3 * ms / max(c, 1):8.1f} us/op {tf} {gb}  {k}")
