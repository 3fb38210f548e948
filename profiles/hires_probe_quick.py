"""UNet evaluation at latent 256^2 (HiresFix shape, CFG batch 2): ms per evaluation."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ldx_amd as ldx
cfg = ldx.UNetConfig.sd15()
eng = ldx.UNetEngine(cfg, ldx.weights.synth_state_dict(ldx.weights.unet_state_dict_spec(cfg), seed=1234), dtype="bf16")
eng.set_graph_mode(True)
lat = int(sys.argv[1]) if len(sys.argv) > 1 else 256
x = torch.randn(2, 4, lat, lat, device="cuda"); sig = torch.full((2,), 5.0, device="cuda"); ctx = torch.randn(2, 77, 768, device="cuda"); out = torch.empty_like(x)
for _ in range(3): eng.denoise(x, sig, ctx, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): eng.denoise(x, sig, ctx, out=out)
e1.record(); torch.cuda.synchronize()
print(f"latent {lat}: {e0.elapsed_time(e1) / 5:.2f} ms per evaluation, launches {eng.plan_info()['launches']}")
