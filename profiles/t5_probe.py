"""Time the full-size T5-XXL encoder (24 blocks, d_model 4096, 64 heads, d_ff 10240; 4.76 B synthetic params) through
ldx_t5_encode for one 256-token prompt (the reference pads T5 prompts to >= 256 tokens, FluxClip.py:593-614)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ldx_amd as ldx

layers = int(sys.argv[1]) if len(sys.argv) > 1 else 24
cfg = ldx.T5Config(num_layers=layers)
t0 = time.perf_counter()
torch.set_num_threads(32)
sd = ldx.weights.synth_state_dict(ldx.weights.t5_state_dict_spec(cfg), seed=1)
print(f"synthetic weights: {time.perf_counter() - t0:.1f} s", flush=True)
t0 = time.perf_counter()
eng = ldx.T5Engine(cfg, sd, dtype="bf16")
del sd
print(f"load + finalize: {time.perf_counter() - t0:.1f} s", flush=True)
for B, L in ((1, 256), (2, 256), (1, 512)):
    ids = torch.randint(0, cfg.vocab_size, (B, L))
    for _ in range(2):
        out = eng.forward(ids)
    torch.cuda.synchronize()
    n = 5
    t0 = time.perf_counter()
    for _ in range(n):
        out = eng.forward(ids)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    info = eng.plan_info()
    print(f"T5-XXL[{layers} blocks] B{B} L{L}: {dt * 1e3:.2f} ms  {info['flops'] / dt / 1e12:.1f} TFLOP/s  launches {info['launches']}  "
          f"weights {4.76e9 * 2 * layers / 24 / dt / 1e12:.2f} TB/s  finite={bool(torch.isfinite(out).all())}")
eng.profile(True)
eng.forward(ids); torch.cuda.synchronize()
eng.profile(False, reset=False)
for k, v in sorted(eng.profile_report().items(), key=lambda kv: -kv[1]["ms"])[:6]:
    print(f"  {k:28s} n={v['count']:3d} {v['ms']:.3f} ms" + (f"  {v['flops'] / v['ms'] / 1e9:.0f} TF" if v['flops'] else ""))
