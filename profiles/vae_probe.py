"""Per-op profile of the VAE decode: where the D = 512 mid-block attention stands.  Usage: python profiles/vae_probe.py [latent=128]"""
import ctypes as C
import json
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ldx_amd as ldx

lat = int(sys.argv[1]) if len(sys.argv) > 1 else 128
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 2          # 2: keyed by kernel class AND op shape
vcfg = ldx.VAEConfig()
vae = ldx.VAEDecoderEngine(vcfg, ldx.weights.synth_state_dict(ldx.weights.vae_decoder_state_dict_spec(vcfg), seed=1, dtype=torch.float32), dtype="bf16")
z = torch.randn(1, 4, lat, lat, device="cuda")
for _ in range(2):
    vae.decode(z)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3):
    vae.decode(z)
e1.record(); torch.cuda.synchronize()
print(f"VAE decode latent {lat}: {e0.elapsed_time(e1) / 3:.2f} ms")
vae._lib.ldx_profile(vae._h, mode, 1)
n = 3
for _ in range(n):
    vae.decode(z)
torch.cuda.synchronize()
vae._lib.ldx_profile(vae._h, 0, 0)
buf = C.create_string_buffer(1 << 18)
vae._lib.ldx_profile_report(vae._h, buf, len(buf))
rep = json.loads(buf.value.decode())
tot = sum(v["ms"] for v in rep.values()) / n
print(f"sum of op times {tot:.3f} ms")
for k, v in sorted(rep.items(), key=lambda kv: -kv[1]["ms"])[:26]:
    ms = v["ms"] / n
    tf = f"{v['flops'] / v['ms'] / 1e9:7.0f} TF" if v["flops"] else ""
    print(f"{ms:8.3f} ms {100 * ms / tot:5.1f}%  n={v['count'] // n:3d}  {tf}  {k}")
