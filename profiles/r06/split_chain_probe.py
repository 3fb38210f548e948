"""Would ONE image's CFG evaluation run faster as two concurrent batch-1 chains (uncond | cond on two streams) than as one batch-2 chain?
And a batch of 8 latents as two concurrent chains of 4?  Two engines (own arenas / graphs), two streams.
Usage: python profiles/r06/split_chain_probe.py [latent=128]"""
import os, sys, time, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ldx_amd as ldx

lat = int(sys.argv[1]) if len(sys.argv) > 1 else 128
cfg = ldx.UNetConfig.sd15()
sd = ldx.weights.synth_state_dict(ldx.weights.unet_state_dict_spec(cfg), seed=1234)
engs = [ldx.UNetEngine(cfg, sd, dtype="bf16") for _ in range(2)]
del sd
for e in engs: e.set_graph_mode(True)
streams = [torch.cuda.Stream() for _ in range(2)]
g = torch.Generator().manual_seed(7)

def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0) / n)
    return statistics.median(ts)

for pb in (1, 8):
    x = torch.randn(pb, 4, lat, lat, generator=g).cuda()
    ctx = torch.randn(2 * pb, 77, 768, generator=g).cuda()
    xx = torch.cat([x, x]).contiguous(); sig = torch.full((2 * pb,), 5.0).cuda()
    out = torch.empty(2 * pb, 4, lat, lat, device="cuda")
    n = 20 if pb == 1 else 5
    t_cfg = bench(lambda: engs[0].denoise_cfg(x, 5.0, ctx, out=out), n)
    engs[0].set_cfg_share(False)
    t_full = bench(lambda: engs[0].denoise_cfg(x, 5.0, ctx, out=out), n)
    engs[0].set_cfg_share(True)
    # (1) uncond | cond as two concurrent half-batch forwards (no sharing): rows [0, pb) and [pb, 2 pb) of the batch
    halves = [(xx[:pb].contiguous(), sig[:pb].contiguous(), ctx[:pb].contiguous(), torch.empty(pb, 4, lat, lat, device="cuda")),
              (xx[pb:].contiguous(), sig[pb:].contiguous(), ctx[pb:].contiguous(), torch.empty(pb, 4, lat, lat, device="cuda"))]
    def split():
        for e, s, h in zip(engs, streams, halves):
            with torch.cuda.stream(s):
                e.denoise(h[0], h[1], h[2], out=h[3])
    for s in streams: s.wait_stream(torch.cuda.current_stream())
    t_split = bench(split, n)
    with torch.cuda.stream(streams[0]):
        t_half = bench(lambda: engs[0].denoise(halves[0][0], halves[0][1], halves[0][2], out=halves[0][3]), n)
    line = f"latent {lat} pb {pb}: batched shared {t_cfg:.3f} ms, batched full {t_full:.3f} ms, one half alone {t_half:.3f} ms, uncond | cond concurrent {t_split:.3f} ms"
    if pb >= 2:
        # (2) the latents as two concurrent chains of pb / 2 (each a CFG evaluation with the shared prefix)
        hb = pb // 2
        parts = [(x[:hb].contiguous(), torch.cat([ctx[:hb], ctx[pb:pb + hb]]).contiguous(), torch.empty(2 * hb, 4, lat, lat, device="cuda")),
                 (x[hb:].contiguous(), torch.cat([ctx[hb:pb], ctx[pb + hb:]]).contiguous(), torch.empty(2 * hb, 4, lat, lat, device="cuda"))]
        def two():
            for e, s, h in zip(engs, streams, parts):
                with torch.cuda.stream(s):
                    e.denoise_cfg(h[0], 5.0, h[1], out=h[2])
        t_two = bench(two, n)
        line += f", two concurrent chains of {hb} latents {t_two:.3f} ms"
    print(line, flush=True)
