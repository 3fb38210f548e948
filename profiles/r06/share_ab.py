"""Same-box A/B of the shared CFG prefix (ldx_unet_cfg_share): the bench's own step loop (CFGDenoiser + fused Euler update, hipGraph) with the prefix shared / not shared.
Usage: python profiles/r06/share_ab.py [steps=20]"""
import os, sys, time, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ldx_amd as ldx

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cfg = ldx.UNetConfig.sd15()
sd = ldx.weights.synth_state_dict(ldx.weights.unet_state_dict_spec(cfg), seed=1234)
eng = ldx.UNetEngine(cfg, sd, dtype="bf16")
eng.set_graph_mode(True)
ms_ = ldx.sampling.ModelSamplingDiscrete()
g = torch.Generator().manual_seed(7)
pos, neg = torch.randn([1, 77, cfg.context_dim], generator=g), torch.randn([1, 77, cfg.context_dim], generator=g)
for (pb, lat, n) in ((1, 128, steps), (1, 64, 2 * steps), (8, 128, max(4, steps // 4))):
    sig = ldx.sampling.calculate_sigmas(ms_, "normal", n + 3)
    x0 = (torch.randn([pb, 4, lat, lat], generator=g) * torch.sqrt(1.0 + sig[0] ** 2.0)).cuda()
    res = {}
    for rnd in range(2):
        for share in (True, False):
            eng.set_cfg_share(share)
            model = ldx.sampling.CFGDenoiser(eng, pos, neg, 7.0, pb, lat, lat)
            x = x0.clone()
            def run(i0, k):
                for i in range(i0, i0 + k):
                    du, dc = model(x, sig[i])
                    ldx.sampling._step(0, x, du, dc, 7.0, sig[i], sig[i + 1] - sig[i])
            run(0, 3); torch.cuda.synchronize()
            ts = []
            for _ in range(3):
                x.copy_(x0); torch.cuda.synchronize(); t0 = time.perf_counter()
                run(3, n); torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0) / n)
            info = eng.plan_info()
            res.setdefault(share, []).append((statistics.median(ts), info, x.clone()))
    a, b = res[True], res[False]
    rel = float((a[0][2].double() - b[0][2].double()).norm() / b[0][2].double().norm())
    print(f"pb {pb} latent {lat}: shared {min(v[0] for v in a):.3f} ms/step ({a[0][1]['launches']} launches, {a[0][1]['flops_executed'] / 1e12:.3f} of {a[0][1]['flops'] / 1e12:.3f} TFLOP executed)"
          f"   full {min(v[0] for v in b):.3f} ms/step ({b[0][1]['launches']} launches)   all rounds shared {[round(v[0], 3) for v in a]} full {[round(v[0], 3) for v in b]}"
          f"   latents after {n} steps rel-L2 shared vs full {rel:.3e}", flush=True)
