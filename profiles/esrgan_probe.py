"""Time the full-size ESRGAN x4 RRDBNet (23 RRDB blocks, 16.7 M params, synthetic weights): one 512x512 tile and a 1024x1024
image through the reference's tiled_scale geometry (tile 512, overlap 32 -> 3x3 tiles, feathered blend) to 4096x4096."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ldx_amd as ldx

cfg = ldx.ESRGANConfig()
sd = ldx.weights.synth_state_dict(ldx.weights.esrgan_state_dict_spec(cfg), seed=3, dtype=torch.float32)
eng = ldx.ESRGANEngine(cfg, sd, dtype="bf16")
x = torch.rand(1, 512, 512, 3, device="cuda")
for _ in range(2): y = eng.forward(x)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 5
for _ in range(n): y = eng.forward(x)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
info = eng.plan_info()
print(f"RRDBNet x4 512^2 -> 2048^2: {dt*1e3:.1f} ms  {info['flops']/dt/1e12:.0f} TFLOP/s ({info['flops']/1e12:.2f} TFLOP incl. channel padding)  launches {info['launches']} "
      f"arena {info['arena_bytes']/2**30:.2f} GiB finite={bool(torch.isfinite(y).all())}")
eng.profile(True); eng.forward(x); torch.cuda.synchronize(); eng.profile(False, reset=False)
for k, v in sorted(eng.profile_report().items(), key=lambda kv: -kv[1]["ms"])[:4]:
    print(f"  {k:30s} n={v['count']:4d} {v['ms']:.2f} ms" + (f"  {v['flops']/v['ms']/1e9:.0f} TF" if v['flops'] else ""))
if len(sys.argv) > 1 and sys.argv[1] == "quick":
    sys.exit(0)
img = torch.rand(1, 1024, 1024, 3)
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = eng.upscale(img)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"ImageUpscaleWithModel 1024^2 -> {tuple(out.shape[1:3])} (9 tiles of 512, overlap 32): {dt*1e3:.0f} ms")
