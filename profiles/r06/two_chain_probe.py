"""How much do kernel-to-kernel tails / ramps cost?  Two INDEPENDENT bs = 1 sampler chains (two engines, two streams, hipGraph each) against one chain:
if the two interleave well, 2 chains take well under 2x one chain — the head-room a cond / uncond split of ONE image's step could tap.
Usage: python profiles/r06/two_chain_probe.py [steps=20] [latent=128]"""
import os, sys, time, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ldx_amd as ldx

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
lat = int(sys.argv[2]) if len(sys.argv) > 2 else 128
cfg = ldx.UNetConfig.sd15()
sd = ldx.weights.synth_state_dict(ldx.weights.unet_state_dict_spec(cfg), seed=1234)
engs = [ldx.UNetEngine(cfg, sd, dtype="bf16") for _ in range(2)]
del sd
ms_ = ldx.sampling.ModelSamplingDiscrete()
sig = ldx.sampling.calculate_sigmas(ms_, "normal", steps + 3)
g = torch.Generator().manual_seed(7)
pos, neg = torch.randn([1, 77, cfg.context_dim], generator=g), torch.randn([1, 77, cfg.context_dim], generator=g)
streams = [torch.cuda.Stream() for _ in range(2)]
chains = []
for e, s in zip(engs, streams):
    e.set_graph_mode(True)
    with torch.cuda.stream(s):
        x = (torch.randn([1, 4, lat, lat], generator=g) * torch.sqrt(1.0 + sig[0] ** 2.0)).cuda()
        model = ldx.sampling.CFGDenoiser(e, pos, neg, 7.0, 1, lat, lat)
    chains.append((e, s, x, model))

def step(c, i):
    e, s, x, model = c
    with torch.cuda.stream(s):
        du, dc = model(x, sig[i])
        ldx.sampling._step(0, x, du, dc, 7.0, sig[i], sig[i + 1] - sig[i])

for c in chains:
    for i in range(3): step(c, i)
torch.cuda.synchronize()
def run(active, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(3, 3 + n):
        for c in active: step(c, i)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n
for _ in range(2):
    one = [run(chains[:1], steps) for _ in range(3)]
    two = [run(chains, steps) for _ in range(3)]
    print(f"latent {lat}: one chain {statistics.median(one):.3f} ms/step; two concurrent chains {statistics.median(two):.3f} ms per step-pair = {statistics.median(two) / 2:.3f} ms per image-step "
          f"({100 * (1 - statistics.median(two) / (2 * statistics.median(one))):.1f} % saved by overlap)", flush=True)
