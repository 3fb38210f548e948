"""bench.py — sampler it/s for SD1.5 1024x1024 bs=1 (BASELINE.json metric) on N MI355X GPUs of one node.

One "step" = one sampler-loop iteration as the reference's tqdm counts it (samplers.py:269): one
CFG-batched UNet evaluation ([uncond; cond] -> batch 2, cond.py:186-226) through ldx_unet_denoise plus the
fused CFG + Euler update (ldx_sampler_step).  Full-size SD1.5 layout (859.5 M params), synthetic seeded
weights and inputs (no checkpoints offline), latent [1,4,128,128], 77-token contexts, cfg 7, sample_euler /
normal schedule, multiscale off = SURVEY.md §8(d) config 2.  Inputs are resident in HBM before timing.

Launching.  `python bench.py --gpus N` with no torchrun environment re-executes itself under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1` (one process per GPU,
backend nccl = RCCL over xGMI); under the driver's own torchrun launch it asserts WORLD_SIZE == --gpus, so a
multi-GPU request can never silently run on one rank.

Workloads.
  --config 2 (default)  one image per GPU: bs = N images in flight, weak scaling, no per-step collective; ONE
                        all-gather of the final latents closes the timed region (north_star: "all-gather of decoded
                        latents only").
  --config 3            SURVEY §8(d) config 3: global batch 64 drawn by ONE prepare_noise(seed) call on the whole
                        batch (ksampler_util.py:274-311), sharded 64/N per GPU (CFG batch 2*64/N), strong scaling.

Timing.  W warm-up steps, then `--repeats` (default 3) timed regions of EXACTLY K steps each, every region bracketed
by barrier + torch.cuda.synchronize() on both sides, MAX over ranks per region, median over regions -> `value`.

Prints ONE JSON line on rank 0 (contract in the task prompt) with extra objects:
  roofline      dominant kernel = the single (kernel, op shape) with the largest share of step time, algorithmic FLOP/s
                from per-op HIP events recorded on the launch stream (ldx_profile mode 2) against the dense bf16 MFMA
                peak (2.5 PFLOP/s, MI355X_MICROARCH.md); HBM traffic per launch from the committed PMC passes
                (profiles/r*/traffic.json); per-class breakdown in `kernels`; whole-step achieved TFLOP/s.
  cpu_baseline  the oracle (CPU restatement, kind "port") timed on this host's cores on the HEADLINE workload
                (1024^2, CFG batch 2, >= 2 evaluations); `reference_cpu` carries the build-container measurement of the
                reference's own path (tests/golden/unet_full.npz, written by oracle/ref_capture_full.py) — never mixed.
  secondary     N = 1 only: the reference-default "euler" name (forced multi-scale, samplers.py:180-184), the end-to-end seconds per
                image (CLIP encode + 20 steps + VAE decode), and SURVEY §8(d) configs 3 / 4 / 5 on the same clock: `config3_shard` (8 latents
                per GPU, CFG batch 16), `flux_fp8` / `flux_bf16` (Flux.1-dev synthetic, one forward and the 28-step sampling chain),
                `hiresfix_2048` (bislerp, 10 x euler_ancestral_cfgpp at latent 256^2, VAE decode 2048^2, ESRGAN tile) — each with its own
                roofline object.  --no-configs skips them (the Flux weights alone take ~1.5 min to build).
"""
import argparse
import hashlib
import json
import os
import socket
import statistics
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0     # dense MFMA bf16, MI355X_MICROARCH.md "Chip-level parameters"


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--repeats", type=int, default=3, help="timed regions of --steps steps each; the median is reported")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--latent", type=int, default=128)
    ap.add_argument("--config", type=int, default=2, choices=(2, 3), help="SURVEY §8(d) config: 2 = one image per GPU (weak), 3 = global batch 64 sharded (strong)")
    ap.add_argument("--global-batch", type=int, default=64, help="config 3 only")
    ap.add_argument("--batch", type=int, default=1, help="config 2 only: images per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the config 3 / 4 / 5 secondary lines (Flux weights take ~1.5 min to build)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--tiny", action="store_true", help="debug: tiny UNet config")
    ap.add_argument("--preflight", action="store_true",
                    help="multi-GPU smoke: init the process group, one 256 KiB all-gather through torch.distributed AND the direct RCCL path, print one JSON "
                         "line (ranks, backend, per-rank device / NUMA node) and exit — seconds, before any weight work")
    ap.add_argument("--no-parity-check", action="store_true", help="skip the 20-step reference-golden comparison after the timed regions")
    ap.add_argument("--private-weights", action="store_true", help="N > 1: every rank synthesises its own state dict (default: local rank 0 publishes one under /dev/shm)")
    ap.add_argument("--stub-engine", action="store_true",
                    help="TEST ONLY (tests/test_bench_gloo.py): CPU + gloo, an analytic per-sample stand-in for the UNet; the line is marked stub")
    return ap.parse_args(argv)


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args):
    """No torchrun environment and N > 1: re-execute under torch.distributed.run, one rank per GPU."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["LDX_RCCL_NONCE"] = f"{os.getpid()}.{time.time_ns()}"          # names this launch: stale hand-off files of earlier runs are ignored (parallel.py)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


# ------------------------------------------------------------------------------------------------------
class StubEngine:
    """TEST ONLY: strictly per-sample analytic denoiser on the CPU (no HIP) so that the N > 1 control flow of this file
    (sharding, timed regions, gather, reductions, JSON) runs under gloo in the CPU test-suite."""
    device = torch.device("cpu")

    def denoise(self, x, sigma, ctx, out=None):
        r = torch.tanh(x) / (1.0 + sigma.view(-1, 1, 1, 1)) + 1e-3 * ctx.mean(dim=(1, 2)).view(-1, 1, 1, 1)
        if out is None:
            return r
        out.copy_(r)
        return out

    def set_graph_mode(self, on):
        pass

    def plan_info(self):
        return {"launches": 0, "flops": 0.0, "arena_bytes": 0}


def stub_step(x, du, dc, cfg, sigma, dt):
    den = du + (dc - du) * cfg
    x += ((x - den) / sigma) * dt


# ------------------------------------------------------------------------------------------------------
def cpu_baseline(cfg, sd, max_eval_s=150.0):
    """Oracle apply_model (CFG batch 2, fp32) on the host cores: one evaluation at 512^2 (config 1 shape), then the
    HEADLINE workload 1024^2: one untimed first touch is NOT spent (too slow); two timed evaluations, mean reported."""
    from oracle import sd15_oracle as O      # baseline leg only — never on the product path
    threads = torch.get_num_threads()
    g = torch.Generator().manual_seed(7)
    ctx = torch.randn([2, 77, cfg.context_dim], generator=g)
    sdf = {k: v.float() for k, v in sd.items()}          # cast once (the reference re-casts per call; excluded here)
    times = {}
    for lat, n in ((64, 1), (128, 2)):
        x = torch.randn([2, 4, lat, lat], generator=g)
        sig = torch.tensor([5.0, 5.0])
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            with torch.no_grad():
                O.apply_model(sdf, cfg, x, sig, ctx)
            ts.append(time.perf_counter() - t0)
            if ts[-1] > max_eval_s:
                break
        times[lat] = ts
    t1024 = sum(times[128]) / len(times[128])
    return {"value": round(1.0 / t1024, 5), "unit": "it/s", "cores": threads, "kind": "port",
            "sample": f"{len(times[128])} CFG-batched UNet evaluations (oracle.apply_model, fp32, batch 2) at 1024x1024 = the headline "
                      f"workload: {', '.join(f'{t:.1f}' for t in times[128])} s each on {threads} torch threads of {os.cpu_count()} host cpus; "
                      f"512x512: {times[64][0]:.2f} s",
            "s_per_it_1024": round(t1024, 2), "s_per_it_512": round(times[64][0], 2)}


def reference_cpu():
    """The reference's OWN path (model.apply_model / KSampler.sample imported from /root/reference), timed in the build
    container when the full-width goldens were captured — carried as data, never re-measured on the GPU box."""
    try:
        import numpy as np
        z = np.load(os.path.join(ROOT, "tests", "golden", "unet_full.npz"))
        t = json.loads(str(z["timing_json"]))
        return {"kind": "reference", "where": "build container (no GPU)", "cpus": t["cpus"], "torch_threads": t["torch_threads"],
                "torch": t["torch"], "s_per_it_512": t.get("ks64_s_per_step"), "s_per_it_1024_first_call": t.get("am128_103_s"),
                "it_per_s_512": round(1.0 / t["ks64_s_per_step"], 4) if t.get("ks64_s_per_step") else None,
                "note": "fp16 weights, fp32 compute (manual_cast), CFG batch 2; 512^2 = mean of 4 sampler steps, 1024^2 = one apply_model call "
                        "(oracle/ref_capture_full.py)"}
    except Exception as e:          # fixture absent: say so instead of inventing a number
        return {"kind": "reference", "error": repr(e)}


def secondary_lines(ldx, unet, cfg, lat, steps):
    """N = 1 extras of SURVEY §8(d): reference-default "euler" (forced multi-scale) and end-to-end seconds per image."""
    vcfg = ldx.VAEConfig()
    vae = ldx.VAEDecoderEngine(vcfg, ldx.weights.synth_state_dict(ldx.weights.vae_decoder_state_dict_spec(vcfg), seed=1, dtype=torch.float32), dtype="bf16")
    ccfg = ldx.CLIPConfig()
    clip = ldx.CLIPTextEngine(ccfg, ldx.weights.synth_state_dict(ldx.weights.clip_state_dict_spec(ccfg), seed=2), dtype="bf16")
    ks = ldx.sampling.KSampler(unet)
    ids = torch.randint(0, 49407, (2, 77), generator=torch.Generator().manual_seed(3))

    def run(sampler, **kw):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        cond = clip.forward(ids, -2)
        cond = cond[0] if isinstance(cond, (tuple, list)) else cond
        pos, neg = cond[0:1].float(), cond[1:2].float()
        torch.cuda.synchronize(); t1 = time.perf_counter()
        trace = []
        x = ks.sample(seed=1, steps=steps, cfg=7.0, sampler_name=sampler, scheduler="normal", positive=pos, negative=neg,
                      latent_image=torch.zeros(1, 4, lat, lat), trace=trace, **kw)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        img = vae.decode(x)
        torch.cuda.synchronize(); t3 = time.perf_counter()
        assert torch.isfinite(img).all()
        return t1 - t0, t2 - t1, t3 - t2, trace

    out = {}
    for name, sampler, kw in (("euler_forced_multiscale", "euler", {}), ("sample_euler_e2e", "sample_euler", dict(enable_multiscale=False))):
        run(sampler, **kw)                                   # warm: plans / graphs for both resolutions
        rs = [run(sampler, **kw) for _ in range(3)]
        samp = statistics.median(r[1] for r in rs)
        tot = statistics.median(r[0] + r[1] + r[2] for r in rs)
        trace = rs[0][3]
        out[name] = {"sampler_name": sampler, "scheduler": "normal", "steps": steps, "it_per_s": round(steps / samp, 3),
                     "unet_evaluations": len(trace), "evaluations_at_half_resolution": sum(1 for s in trace if s[0] != lat),
                     "clip_ms": round(1e3 * statistics.median(r[0] for r in rs), 3), "sampler_ms": round(1e3 * samp, 2),
                     "vae_decode_ms": round(1e3 * statistics.median(r[2] for r in rs), 2), "e2e_s_per_image": round(tot, 4),
                     "runs": 3}
    return out


def _class_table(rep, nrun):
    cls = {}
    for k, v in rep.items():
        e = cls.setdefault(k.split(" ")[0], {"count": 0, "ms": 0.0, "flops": 0.0})
        e["count"] += v["count"]; e["ms"] += v["ms"]; e["flops"] += v["flops"]
    return {k: {"launches": v["count"] // nrun, "ms": round(v["ms"] / nrun, 3), **({"tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1)} if v["flops"] > 0 and v["ms"] > 0 else {})}
            for k, v in sorted(cls.items(), key=lambda kv: -kv[1]["ms"])[:5]}


def _exec_flops(info):
    """FLOPs the plan EXECUTES (ldx_plan_flops): with the shared CFG prefix (ldx_unet_cfg_share, default on) a CFG evaluation computes everything in front of the
    first cross-attention once for both halves, so this is smaller than the algorithmic count `flops` (the reference's arithmetic); roofline fractions use it."""
    return info.get("flops_executed", info["flops"])


def _roof(flops, ms, peak, unit_note):
    tf = flops / (ms * 1e-3) / 1e12
    return {"bound": "mfma", "achieved": round(tf, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(tf / peak, 4), "traffic": None, "of": unit_note}


def sd15_512_line(ldx, eng, cfg, steps):
    """BASELINE config 1's shape on the GPU engine (SD1.5 512 x 512, latent 64^2, CFG batch 2; the reference runs that config on its CPU path): same loop as the
    headline at a quarter of the pixels, where the step is a flat profile of 10-20 us kernels."""
    lat = 64
    ms_ = ldx.sampling.ModelSamplingDiscrete()
    sig = ldx.sampling.calculate_sigmas(ms_, "normal", steps + 2)
    g = torch.Generator().manual_seed(13)
    pos, neg = torch.randn([1, 77, cfg.context_dim], generator=g), torch.randn([1, 77, cfg.context_dim], generator=g)
    x = (torch.randn([1, 4, lat, lat], generator=g) * torch.sqrt(1.0 + sig[0] ** 2.0)).cuda()
    model = ldx.sampling.CFGDenoiser(eng, pos, neg, 7.0, 1, lat, lat)

    def run(i0, n):
        for i in range(i0, i0 + n):
            du, dc = model(x, sig[i])
            ldx.sampling._step(0, x, du, dc, 7.0, sig[i], sig[i + 1] - sig[i])
    run(0, 2); torch.cuda.synchronize()
    x0 = x.clone(); ts = []
    for _ in range(3):
        x.copy_(x0); torch.cuda.synchronize(); t0 = time.perf_counter()
        run(2, steps); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / steps)
    ms = 1e3 * statistics.median(ts)
    info = eng.plan_info()
    assert torch.isfinite(x).all()
    return {"workload": f"SD1.5 512^2 bs=1 (latent 64^2, CFG batch 2), sample_euler/normal, {steps} steps x 3 regions (median)",
            "ms_per_step": round(ms, 3), "it_per_s": round(1e3 / ms, 2), "step_tflop": round(info["flops"] / 1e12, 3), "step_tflop_executed": round(_exec_flops(info) / 1e12, 3),
            "launches_per_step": info["launches"],
            "roofline": _roof(_exec_flops(info), ms, PEAK_BF16_TFLOPS, "whole step (executed flops), dense bf16 MFMA peak")}


def config3_shard_line(ldx, eng, cfg, lat, steps):
    """SURVEY config 3's PER-GPU shard on one GPU: 8 latents, CFG batch 16, same loop as the headline (pipeline shape bs = 64 over 8 GPUs)."""
    pb = 8
    ms_ = ldx.sampling.ModelSamplingDiscrete()
    sig = ldx.sampling.calculate_sigmas(ms_, "normal", steps + 2)
    g = torch.Generator().manual_seed(11)
    pos, neg = torch.randn([1, 77, cfg.context_dim], generator=g), torch.randn([1, 77, cfg.context_dim], generator=g)
    x = (torch.randn([pb, 4, lat, lat], generator=g) * torch.sqrt(1.0 + sig[0] ** 2.0)).cuda()
    model = ldx.sampling.CFGDenoiser(eng, pos, neg, 7.0, pb, lat, lat)

    def run(i0, n):
        for i in range(i0, i0 + n):
            du, dc = model(x, sig[i])
            ldx.sampling._step(0, x, du, dc, 7.0, sig[i], sig[i + 1] - sig[i])
    run(0, 2); torch.cuda.synchronize()
    x0 = x.clone(); ts = []
    for _ in range(3):
        x.copy_(x0); torch.cuda.synchronize(); t0 = time.perf_counter()
        run(2, steps); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / steps)
    ms = 1e3 * statistics.median(ts)
    info = eng.plan_info()
    assert torch.isfinite(x).all()
    return {"workload": f"SD1.5 1024^2, {pb} latents per GPU (CFG batch {2 * pb}) = the per-GPU share of bs 64 over 8 GPUs, sample_euler/normal, {steps} steps x 3 regions (median)",
            "ms_per_step": round(ms, 2), "image_steps_per_s": round(pb * 1e3 / ms, 2), "step_tflop": round(info["flops"] / 1e12, 2), "step_tflop_executed": round(_exec_flops(info) / 1e12, 2),
            "roofline": _roof(_exec_flops(info), ms, PEAK_BF16_TFLOPS, "whole step (executed flops), dense bf16 MFMA peak")}


def flux_lines(ldx, steps=28):
    """BASELINE config 4: Flux.1-dev DiT (19 + 38 blocks, 11.9 B synthetic parameters), 1024^2 (4096 image + 256 text tokens), the reference's
    Flux sampling chain (pipeline.py:215-277: euler_cfgpp / beta, cfg 1 with a zeroed negative -> both branches evaluated, batch 2, guidance 3.0)
    for 28 steps, in the MX fp8 mode and in bf16."""
    cfg = ldx.FluxConfig()
    spec = ldx.weights.flux_state_dict_spec(cfg)
    g = torch.Generator().manual_seed(1)
    sd = {}
    for k, shp in spec:        # cheap fill: one random block tiled (values do not matter for timing; still random data)
        n = 1
        for d in shp:
            n *= d
        if k.endswith(".bias"):
            sd[k] = (0.02 * torch.randn(shp, generator=g)).half()
        elif k.endswith("scale"):
            sd[k] = torch.ones(shp).half()
        else:
            base = torch.randn(min(n, 1 << 20), generator=g) / (shp[-1] ** 0.5)
            sd[k] = base.repeat((n + base.numel() - 1) // base.numel())[:n].reshape(shp).half()
    out = {}
    pos = (torch.randn(1, 256, 4096, generator=g), torch.randn(1, 768, generator=g))
    neg = (torch.zeros(1, 256, 4096), torch.zeros(1, 768))
    x1 = torch.randn(1, 16, 128, 128, device="cuda"); t1 = torch.tensor([0.7], device="cuda"); gd = torch.tensor([3.0], device="cuda")
    for name, fp8, peak in (("flux_fp8", True, 5000.0), ("flux_bf16", False, PEAK_BF16_TFLOPS)):
        eng = ldx.FluxEngine(cfg, sd, dtype="bf16", fp8=("attn" if fp8 else False))      # config 4 "fp8 MFMA": the explicit full mode (ldx_flux_set_fp8 3)
        for _ in range(2):
            eng.denoise(x1, t1, pos[0].cuda(), pos[1].cuda(), gd)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            o = eng.denoise(x1, t1, pos[0].cuda(), pos[1].cuda(), gd)
        torch.cuda.synchronize(); fwd_ms = 1e3 * (time.perf_counter() - t0) / 5
        info1 = eng.plan_info()
        eng.profile(True); eng.denoise(x1, t1, pos[0].cuda(), pos[1].cuda(), gd); torch.cuda.synchronize(); eng.profile(False, reset=False)
        kern = _class_table(eng.profile_report(), 1)
        ks = ldx.sampling.FluxKSampler(eng)
        ts = []
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            lat = ks.sample(seed=1, steps=steps, cfg=1, sampler_name="euler_cfgpp", scheduler="beta", positive=pos, negative=neg,
                            latent_image=torch.zeros(1, 16, 128, 128), guidance=3.0)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        assert torch.isfinite(lat).all() and torch.isfinite(o).all()
        out[name] = {"workload": f"Flux.1-dev DiT 19+38 blocks (11.9 B synthetic params), 1024^2 = 4096 image + 256 text tokens, {'MX fp8 (e4m3 + E8M0 per 32) linears AND attention (QK^T / PV on the block-scaled MFMA, csrc/attn_mx.hip)' if fp8 else 'bf16 linears and attention'}",
                     "ms_per_forward_bs1": round(fwd_ms, 2), "forward_tflop": round(info1["flops"] / 1e12, 2),
                     "sampler": f"FluxKSampler euler_cfgpp/beta, {steps} steps, cfg 1 with a zeroed negative (batch-2 evaluations, second run of 2)",
                     "sampler_s": round(ts[-1], 3), "it_per_s": round(steps / ts[-1], 3), "launches": info1["launches"], "kernels": kern,
                     "roofline": _roof(info1["flops"], fwd_ms, peak, f"one batch-1 forward, dense {'fp8' if fp8 else 'bf16'} MFMA peak")}
        eng.close(); del eng, ks
        torch.cuda.empty_cache()
    return out


def hiresfix_line(ldx, unet, cfg):
    """BASELINE config 5 (pipeline.py:346-366): 1024^2 latents -> LatentUpscale bislerp x2 -> 10 steps euler_ancestral_cfgpp / normal at denoise 0.45 on the
    256^2 latent (CFG batch 2) -> VAE decode 2048^2 (untiled) -> one 512^2 tile through the ESRGAN x4 RRDBNet (the UltimateSDUpscale tile size)."""
    vcfg = ldx.VAEConfig()
    vae = ldx.VAEDecoderEngine(vcfg, ldx.weights.synth_state_dict(ldx.weights.vae_decoder_state_dict_spec(vcfg), seed=1, dtype=torch.float32), dtype="bf16")
    ecfg = ldx.ESRGANConfig()
    esr = ldx.ESRGANEngine(ecfg, ldx.weights.synth_state_dict(ldx.weights.esrgan_state_dict_spec(ecfg), seed=3, dtype=torch.float32), dtype="bf16")
    ks = ldx.sampling.KSampler(unet)
    g = torch.Generator().manual_seed(5)
    pos, neg = torch.randn([1, 77, cfg.context_dim], generator=g), torch.randn([1, 77, cfg.context_dim], generator=g)
    base = torch.randn([1, 4, 128, 128], generator=g).cuda()
    tile = torch.rand(1, 512, 512, 3, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    rs = []
    # GPU-event time of every UNet evaluation inside the sampler loop: the loop's wall clock also carries euler_ancestral's noise draws from the CPU RNG
    # (the reference's stream) and their uploads, 2-20 ms per step depending on the host, which is not what "ms per evaluation" is meant to say
    evs = []
    def _timed(fn):
        def w(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); r = fn(*a, **k); e1.record(); evs.append((e0, e1)); return r
        return w
    orig = {n: getattr(unet, n) for n in ("denoise_cfg", "denoise") if hasattr(unet, n)}
    for n, f in orig.items():
        setattr(unet, n, _timed(f))
    ev_gpu = []
    for rep in range(3):
        evs.clear()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        up = ldx.latent_upscale(base, 2048, 2048)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        trace = []
        hi = ks.sample(seed=2, steps=10, cfg=8.0, denoise=0.45, sampler_name="euler_ancestral_cfgpp", scheduler="normal", positive=pos, negative=neg,
                       latent_image=up, trace=trace)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        img = vae.decode(hi)
        torch.cuda.synchronize(); t3 = time.perf_counter()
        big = esr.forward(tile)
        torch.cuda.synchronize(); t4 = time.perf_counter()
        rs.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3, len(trace)))
        ev_gpu.append(statistics.median(a.elapsed_time(b) for a, b in evs) if evs else float("nan"))
    for n, f in orig.items():
        setattr(unet, n, f)
    assert torch.isfinite(img).all() and torch.isfinite(big).all()
    med = lambda i: statistics.median(r[i] for r in rs[1:])
    nev = rs[-1][4]
    uinfo, vinfo, einfo = unet.plan_info(), vae.plan_info(), esr.plan_info()
    ev_ms = statistics.median(ev_gpu[1:])                  # median evaluation (GPU events) of the warm repetitions
    ev_wall_ms = 1e3 * med(1) / max(nev, 1)
    return {"workload": "SD1.5 HiresFix 2048^2: bislerp 128^2 -> 256^2 latent, 10 steps euler_ancestral_cfgpp/normal denoise 0.45 (CFG batch 2), VAE decode 2048^2 untiled, ESRGAN x4 on one 512^2 tile",
            "bislerp_ms": round(1e3 * med(0), 2), "sampler_ms": round(1e3 * med(1), 1), "unet_evaluations": nev, "ms_per_evaluation": round(ev_ms, 2), "ms_per_evaluation_wall": round(ev_wall_ms, 2),
            "vae_decode_2048_ms": round(1e3 * med(2), 1), "esrgan_tile_ms": round(1e3 * med(3), 1), "total_s": round(med(0) + med(1) + med(2) + med(3), 3),
            "vae_arena_gib": round(vinfo["arena_bytes"] / 2 ** 30, 2),
            "roofline": _roof(_exec_flops(uinfo), ev_ms, PEAK_BF16_TFLOPS, f"one UNet evaluation at latent 256^2 ({uinfo['flops'] / 1e12:.1f} TFLOP algorithmic, {_exec_flops(uinfo) / 1e12:.1f} executed), dense bf16 MFMA peak"),
            "vae_roofline": _roof(vinfo["flops"], 1e3 * med(2), PEAK_BF16_TFLOPS, "VAE decode 2048^2"),
            "esrgan_roofline": _roof(einfo["flops"], 1e3 * med(3), PEAK_BF16_TFLOPS, "RRDBNet x4, 512^2 tile (flops incl. channel padding)")}


def run_preflight(ldx, dist, rank, world, local_rank, dev, stub):
    """A topology failure must cost seconds and be NAMED: one 256 KiB all-gather through torch.distributed (asserted) and, on GPUs, the same through
    the direct RCCL path (guarded, reported), plus every rank's device / NUMA node / host cpus.  Runs before any weight work."""
    t0 = time.perf_counter()
    n = 65536                                                     # 256 KiB of fp32 per rank
    mine = torch.full((n,), float(rank), dtype=torch.float32, device=dev)
    chunks = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(chunks, mine)
    if not stub:
        torch.cuda.synchronize()
    for r, c in enumerate(chunks):
        assert float(c[0]) == float(r) and float(c[-1]) == float(r), f"preflight all-gather: chunk {r} holds {float(c[0])}"
    t_pg = time.perf_counter() - t0
    info = {"rank": rank, "local_rank": local_rank, "host": socket.gethostname(), "pid": os.getpid(),
            "device": (None if stub else torch.cuda.get_device_name(local_rank)),
            "numa_node": (None if stub else ldx.parallel.gpu_numa_node(local_rank)),
            "cpus_allowed": len(os.sched_getaffinity(0)), "visible_devices": os.environ.get("HIP_VISIBLE_DEVICES")}
    infos = [None] * world
    dist.all_gather_object(infos, info)
    direct = None
    if not stub:
        try:
            def _xchg(r, make_id):
                box = [make_id() if r == 0 else None]
                dist.broadcast_object_list(box, src=0)
                return box[0]
            td0 = time.perf_counter()
            comm = ldx.parallel.RcclComm(rank, world, id_exchange=_xchg)
            got = comm.all_gather_latents(mine.view(1, n), world)
            torch.cuda.synchronize()
            ok = all(float(got[r, 0]) == float(r) for r in range(world))
            comm.close()
            direct = {"ok": bool(ok), "s": round(time.perf_counter() - td0, 3)}
        except Exception as e:      # noqa: BLE001 — reported in the line
            direct = {"ok": False, "error": repr(e)[:300]}
    return {"world": world, "backend": dist.get_backend(), "allgather_256KiB_ok": True, "process_group_s": round(t_pg, 3),
            "rccl_direct": direct, "ranks": infos}


def parity_check(ldx, eng, dtype):
    """Did the timed loop time the RIGHT work?  Runs BASELINE config 2 exactly as the reference golden was captured (oracle/ref_capture_full20.py:
    KSampler.sample, sample_euler / normal, 20 steps, cfg 7, seed 42, multiscale off, zero latent 128^2, the fixture's own prompt tensors) through the
    same engine object the timed regions used, and compares with the reference's latents.  Fixture only: the oracle stays out of this."""
    import numpy as np
    path = os.path.join(ROOT, "tests", "golden", "unet_full20.npz")
    try:
        z = np.load(path)
        want = torch.from_numpy(z["ks128_20_out"]).double()
        P, N = torch.from_numpy(z["P"]), torch.from_numpy(z["N"])
    except Exception as e:          # fixture absent: say so instead of inventing a number
        return {"ok": False, "error": repr(e)[:200]}
    tol = {"bf16": 3e-2, "f16": 5e-3, "fp16": 5e-3}[dtype]          # measured 1.8e-2 / 2.0e-3 (tests/test_fullwidth_gpu.py)
    ks = ldx.sampling.KSampler(eng)
    got = ks.sample(seed=42, steps=20, cfg=7.0, sampler_name="sample_euler", scheduler="normal", enable_multiscale=False,
                    positive=P, negative=N, latent_image=torch.zeros(1, 4, 128, 128)).double().cpu()
    rel = float((got - want).norm() / want.norm())
    cos = float(torch.dot(got.flatten(), want.flatten()) / (got.norm() * want.norm()))
    return {"rel_l2": round(rel, 6), "cos": round(cos, 7), "tol": tol, "cos_min": 0.999, "ok": bool(rel <= tol and cos >= 0.999),
            "against": "tests/golden/unet_full20.npz ks128_20_out = the reference's own KSampler.sample latents (20 steps, 1024x1024, cfg 7, seed 42)",
            "latents_sha256_16": hashlib.sha256(got.float().contiguous().numpy().tobytes()).hexdigest()[:16]}


# ------------------------------------------------------------------------------------------------------
def main(argv=None):
    args = parse_args(argv)
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(self_launch(args))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"bench.py --gpus {args.gpus} is running with WORLD_SIZE={world}: refusing to mis-report n_gpus"
    stub = args.stub_engine
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if stub:
            dist_mod.init_process_group("gloo", rank=rank, world_size=world)
        else:
            torch.cuda.set_device(local_rank)
            dist_mod.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        dist = dist_mod
        assert dist.get_world_size() == args.gpus
    if stub:
        dev = torch.device("cpu")
        sync = lambda: None
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        sync = torch.cuda.synchronize

    import ldx_amd as ldx
    t_start = time.perf_counter()
    preflight = None
    if world > 1:
        preflight = run_preflight(ldx, dist, rank, world, local_rank, dev, stub)      # before any weight work
    if args.preflight:
        if preflight is None:
            preflight = {"world": 1, "backend": None, "ranks": [{"rank": 0, "device": (None if stub else torch.cuda.get_device_name(local_rank)),
                                                                  "numa_node": (None if stub else ldx.parallel.gpu_numa_node(local_rank))}]}
        if rank == 0:
            print(json.dumps({"preflight": preflight, "n_gpus": world}), flush=True)
        if dist:
            dist.barrier()
            dist.destroy_process_group()
        return
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if world > 1:
        # N ranks convert the same 859.5 M parameters at once: each gets the cores of its GPU's NUMA node (its share of them), not N x all
        cpus = ldx.parallel.bind_rank_to_numa(local_rank, local_world) if not stub else list(range(max(1, (os.cpu_count() or 1) // world)))
        torch.set_num_threads(max(1, len(cpus)))
    cfg = ldx.UNetConfig.tiny(64, 128) if args.tiny else ldx.UNetConfig.sd15()
    sd = None
    build = {}
    if stub:
        eng = StubEngine()
    else:
        spec = ldx.weights.unet_state_dict_spec(cfg)
        tb0 = time.perf_counter()
        if world > 1 and not args.private_weights:
            # ONE synthesis per node (parallel.shared_state_dict: free-space check, agreed fall-back to private synthesis, tests/test_parallel_gloo.py)
            sd, build["weights"] = ldx.parallel.shared_state_dict(dist, spec, rank, world, local_rank, local_world, seed=1234, tag=os.environ.get("MASTER_PORT", "0"))
        else:
            sd = ldx.weights.synth_state_dict(spec, seed=1234)
            build["weights"] = "private synthesis"
        tb1 = time.perf_counter()
        eng = ldx.UNetEngine(cfg, sd, device=local_rank, dtype=args.dtype)
        build.update(state_dict_s=round(tb1 - tb0, 2), engine_build_s=round(time.perf_counter() - tb1, 2), since_start_s=round(time.perf_counter() - t_start, 2))
        if not args.no_graph:
            eng.set_graph_mode(True)
    if dist:
        builds = [None] * world
        dist.all_gather_object(builds, build)
    else:
        builds = [build]

    lat = args.latent
    if args.config == 3:
        gb = args.global_batch
        lo_, hi_ = ldx.parallel.shard_bounds(gb, rank, world)          # uneven batches: the remainder goes to the first ranks
        pb, scaling = hi_ - lo_, "strong"
        assert pb > 0, f"config 3: global batch {gb} leaves rank {rank} of {world} without work"
    else:
        pb, gb, scaling = args.batch, world * args.batch, "weak"
    total = args.warmup + args.steps
    ms = ldx.sampling.ModelSamplingDiscrete()
    sigmas = ldx.sampling.calculate_sigmas(ms, "normal", total)
    g = torch.Generator().manual_seed(7)
    pos = torch.randn([1, 77, cfg.context_dim], generator=g)
    neg = torch.randn([1, 77, cfg.context_dim], generator=g)
    # ONE draw for the whole global batch (reference semantics), then this rank's contiguous slice
    noise = ldx.parallel.shard_noise((gb, 4, lat, lat), 42, rank, world)
    assert noise.shape[0] == pb
    x = (noise * torch.sqrt(1.0 + sigmas[0] ** 2.0)).to(dev)
    model = ldx.sampling.CFGDenoiser(eng, pos, neg, 7.0, pb, lat, lat)

    def run_steps(i0, n):
        for i in range(i0, i0 + n):
            du, dc = model(x, sigmas[i])
            if stub:
                stub_step(x, du, dc, 7.0, sigmas[i], sigmas[i + 1] - sigmas[i])
            else:
                ldx.sampling._step(0, x, du, dc, 7.0, sigmas[i], sigmas[i + 1] - sigmas[i])

    def fence():
        sync()
        if dist:
            dist.barrier()
        sync()

    run_steps(0, args.warmup)
    sync()
    x_start = x.clone()
    regions, gathers, gpu_ms = [], [], []
    gathered = None
    for rep in range(max(1, args.repeats)):
        x.copy_(x_start)
        fence()
        if not stub:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        t0 = time.perf_counter()
        run_steps(args.warmup, args.steps)
        if not stub:
            ev1.record()
        sync()
        tg0 = time.perf_counter()
        gathered = ldx.parallel.gather_latents(x, gb, dist)    # final latents only: the job's single collective
        sync()
        tg1 = time.perf_counter()
        fence()
        regions.append(time.perf_counter() - t0)
        gathers.append(tg1 - tg0)
        if not stub:
            gpu_ms.append(ev0.elapsed_time(ev1) / args.steps)
    assert gathered.shape[0] == gb, (gathered.shape, gb)
    assert torch.isfinite(gathered).all(), "non-finite latents"

    # The same collective once more through the DIRECT RCCL path (ctypes on librccl.so: ncclCommInitRank + ncclAllGather on this stream; the unique id
    # travels through the process group's store, not a data-path collective).  Outside the timed regions and guarded: the first multi-GPU run of this
    # code must not lose its headline to an untested call; the line reports whether it ran, its time, and that it returned the same bytes.
    rccl_direct = None
    if world > 1 and not stub:
        try:
            def _xchg(r, make_id):
                box = [make_id() if r == 0 else None]
                dist.broadcast_object_list(box, src=0)
                return box[0]
            comm = ldx.parallel.RcclComm(rank, world, id_exchange=_xchg)
            comm.all_gather_latents(x, gb); sync()                                   # warm
            fence(); td0 = time.perf_counter()
            direct = comm.all_gather_latents(x, gb); sync()
            td1 = time.perf_counter()
            rccl_direct = {"ok": True, "ms": round(1e3 * (td1 - td0), 3), "equal_to_torch_distributed": bool(torch.equal(direct, gathered))}
            comm.close()
        except Exception as e:      # noqa: BLE001 — reported in the line
            rccl_direct = {"ok": False, "error": repr(e)[:300]}

    # config 3, end to end: every rank decodes ITS OWN shard of the final latents (the decode is sharded like the sampling; gathering u8 RGB would move
    # 3 MiB per image against 64 KiB per latent, so the images stay with the rank that made them and only the latents are gathered).  After the timed
    # regions, guarded like the direct RCCL check: reported next to the headline, never part of it.
    e2e3 = None
    if args.config == 3 and not stub and not args.tiny:
        try:
            vcfg = ldx.VAEConfig()
            vae = ldx.VAEDecoderEngine(vcfg, ldx.weights.synth_state_dict(ldx.weights.vae_decoder_state_dict_spec(vcfg), seed=1, dtype=torch.float32), device=local_rank, dtype="bf16")
            mine_lat = gathered[lo_:hi_].contiguous()
            vae.decode(mine_lat[:1]); sync()                                        # plan + warm
            fence(); tv0 = time.perf_counter()
            for i in range(pb):
                img = vae.decode(mine_lat[i:i + 1])
            sync(); tv1 = time.perf_counter()
            assert torch.isfinite(img.float()).all()
            dec_ms = 1e3 * (tv1 - tv0)
            if dist:
                tmx = torch.tensor([dec_ms], device=dev, dtype=torch.float64)
                dist.all_reduce(tmx, op=dist.ReduceOp.MAX)
                dec_ms = float(tmx.item())
            e2e3 = {"ok": True, "vae_decodes_per_rank": pb, "vae_decode_ms_per_rank_max": round(dec_ms, 2), "where": "each rank decodes its own shard after the latent gather"}
            del vae
        except Exception as e:      # noqa: BLE001 — reported in the line
            e2e3 = {"ok": False, "error": repr(e)[:300]}

    # MAX over ranks per region; per-rank medians for the report
    per_rank_ms = [1000.0 * statistics.median(regions) / args.steps]
    if dist:
        t = torch.tensor(regions + gathers, device=dev, dtype=torch.float64)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        n = len(regions)
        regions_max, gathers_max = tmax[:n].tolist(), tmax[n:].tolist()
        mine = torch.tensor([per_rank_ms[0]], device=dev, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank_ms = [float(v.item()) for v in allr]
    else:
        regions_max, gathers_max = regions, gathers
    elapsed = statistics.median(regions_max)
    # every rank must hold the same gathered batch: fold it into a checksum and compare across ranks
    if dist:
        cs = gathered.double().sum().reshape(1).to(dev)
        lo, hi = cs.clone(), cs.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        assert float(lo.item()) == float(hi.item()), "ranks disagree on the gathered latents"

    info = eng.plan_info()
    # ---- per-kernel roofline leg: same process, same data, HIP events per op on the launch stream ----
    roof = None
    if rank == 0 and not stub:
        eng.set_graph_mode(False)
        eng._lib.ldx_profile(eng._h, 2, 1)            # mode 2: keyed by kernel class AND op shape
        xs = x_start.clone()
        nprof = 3
        for i in range(nprof):
            du, dc = model(xs, sigmas[args.warmup])
        torch.cuda.synchronize()
        eng.profile(False, reset=False)
        rep = eng.profile_report()
        if not args.no_graph:
            eng.set_graph_mode(True)
        tot_ms = sum(v["ms"] for v in rep.values())
        cls = {}
        for k, v in rep.items():
            c = k.split(" ")[0]
            e = cls.setdefault(c, {"count": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
            for f in e:
                e[f] += v[f]
        kern = {}
        for k, v in sorted(cls.items(), key=lambda kv: -kv[1]["ms"]):
            e = {"launches_per_step": v["count"] // nprof, "ms_per_step": round(v["ms"] / nprof, 4), "share": round(v["ms"] / tot_ms, 4)}
            if v["flops"] > 0:
                e["tflops"] = round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2)
            if v["bytes"] > 0 and v["ms"] > 0:
                e["alg_GBps"] = round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1)
            kern[k] = e
        # the dominant KERNEL = the single (kernel, shape) with the most time: one launch geometry, so that "per launch"
        # flops, duration and PMC traffic all refer to the same thing
        dom = max((k for k in rep if rep[k]["flops"] > 0), key=lambda k: rep[k]["ms"])
        dv = rep[dom]
        dom_tflops = dv["flops"] / (dv["ms"] * 1e-3) / 1e12
        traffic, traffic_src, device_kernel = None, None, None
        try:
            import glob
            tj = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "traffic.json")))[-1]
            raw = open(tj, "rb").read()
            tjd = json.loads(raw)
            traffic = tjd.get(dom, {}).get("bytes")      # HBM bytes per launch from the committed PMC passes (rocprofv3 --pmc cannot run inside this process)
            device_kernel = tjd.get(dom, {}).get("device_kernel")      # the mangled name rocprofv3 reports for this launch (profiles/r*/bench_kernel_stats.csv)
            traffic_src = {"file": os.path.relpath(tj, ROOT), "git_blob_sha1": hashlib.sha1(b"blob %d\0" % len(raw) + raw).hexdigest(),
                           "kernel_in_file": dom in tjd, "collected_for": tjd.get("_meta", {}).get("kernel_build")}
        except Exception:
            traffic = None
        step_ms_gpu = statistics.median(gpu_ms)
        roof = {"bound": "mfma", "kernel": dom, "device_kernel": device_kernel, "achieved": round(dom_tflops, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                "frac": round(dom_tflops / PEAK_BF16_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "launches_per_step": dv["count"] // nprof, "avg_launch_ms": round(dv["ms"] / dv["count"], 4),
                "flop_per_launch": round(dv["flops"] / dv["count"] / 1e9, 2),
                "share_of_step": round(dv["ms"] / tot_ms, 4),
                "step_tflop": round(info["flops"] / 1e12, 4), "step_tflop_executed": round(_exec_flops(info) / 1e12, 4),
                "step_tflop_note": "step_tflop = the reference's arithmetic for one CFG evaluation (SURVEY §8d); executed = what the plan runs: everything in front of the first "
                                   "cross-attention is identical in both CFG halves and computed once (ldx_unet_cfg_share); step_achieved / step_frac use the executed count",
                "step_ms": round(step_ms_gpu, 3),
                "step_achieved": round(_exec_flops(info) / (step_ms_gpu * 1e-3) / 1e12, 2),
                "step_frac": round(_exec_flops(info) / (step_ms_gpu * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
                "kernels": kern}

    secondary = None
    if rank == 0 and world == 1 and not stub and not args.no_secondary and not args.tiny and args.config == 2 and pb == 1:
        secondary = secondary_lines(ldx, eng, cfg, lat, args.steps)
        if lat == 128 and not args.no_configs:
            # SURVEY §8(d) configs 3, 4, 5 on the driver's clock (each guarded: a failure is reported in the line, the headline stays valid)
            for key, fn in (("sd15_512", lambda: sd15_512_line(ldx, eng, cfg, 40)),
                            ("config3_shard", lambda: config3_shard_line(ldx, eng, cfg, lat, 10)),
                            ("hiresfix_2048", lambda: hiresfix_line(ldx, eng, cfg)),
                            ("flux", lambda: flux_lines(ldx))):
                t0 = time.perf_counter()
                try:
                    r = fn()
                except Exception as e:          # noqa: BLE001 — reported, not swallowed
                    r = {"error": repr(e)}
                if key == "flux" and "error" not in r:
                    for k2, v2 in r.items():
                        secondary[k2] = v2
                    secondary["flux_wall_s"] = round(time.perf_counter() - t0, 1)
                else:
                    r["wall_s"] = round(time.perf_counter() - t0, 1)
                    secondary[key] = r
                torch.cuda.empty_cache()

    # ---- did the timed regions time the right work?  (N = 1, headline engine, after every timed region; fixture comparison only) ----
    parity = None
    if rank == 0 and world == 1 and not stub and not args.tiny and not args.no_parity_check:
        try:
            parity = parity_check(ldx, eng, args.dtype)
        except Exception as e:      # noqa: BLE001 — reported in the line
            parity = {"ok": False, "error": repr(e)[:300]}

    cpu = None
    if rank == 0 and world == 1 and not stub and not args.no_cpu_baseline and not args.tiny:
        cpu = cpu_baseline(cfg, sd)
        cpu["reference_cpu"] = reference_cpu()

    if rank == 0:
        # config 2: N independent images advance one sampler iteration per step -> iterations/s summed over ranks (weak);
        # config 3: the whole bs = 64 job advances one iteration per step (strong)
        its = (world if args.config == 2 else 1) * args.steps / elapsed
        headline = (args.config == 2 and pb == 1 and lat == 128 and not args.tiny and not stub)
        line = {
            "metric": ("STUB " if stub else "") + (f"sampler it/s (UNet steps/sec) SD1.5 {lat * 8}x{lat * 8} bs={pb if args.config == 2 else gb} {args.dtype}"
                                                   + ("" if args.config == 2 else f" (SURVEY config 3: global batch {gb} sharded over {world} GPU(s))")),
            "value": round(its, 3), "unit": "it/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1000.0 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": scaling,
            "vs_baseline": (round(its / 2.8, 3) if (world == 1 and headline) else None), "dtype": args.dtype,
            "data": "stub (CPU test of the harness)" if stub else "synthetic",
            "config": {"workload": f"SD1.5 UNet (859.5M params, synthetic seeded weights) sampler loop, latent "
                                   f"[{pb},4,{lat},{lat}] ({lat * 8}x{lat * 8}) per GPU, CFG batch {2 * pb}, ctx 77x768, sample_euler/normal, "
                                   f"multiscale off; SURVEY config {args.config}: {gb} image(s) in flight over {world} GPU(s)",
                       "survey_config": args.config, "global_batch": gb, "images_per_gpu": (pb if args.config == 2 else -(-gb // world)),
                       "image_steps_per_s": round(gb * args.steps / elapsed, 3),
                       "parallelism": f"batch-shard x{world} (replicated weights, one final all-gather of latents)",
                       "launches_per_step": info["launches"], "hip_graph": not args.no_graph,
                       "vs_baseline_note": "2.8 it/s = README table, RTX 3060 mobile + Stable-Fast (BASELINE.md §1)"},
            "timing": {"repeats": len(regions_max), "statistic": "median of regions; each region = max over ranks",
                       "wall_region": "K steps + the final all-gather of the latents + the closing fence (ms_per_step, value); roofline.step_ms is the GPU-event time of the K steps alone",
                       "region_ms_per_step": [round(1000.0 * r / args.steps, 3) for r in regions_max],
                       "per_rank_ms_per_step": [round(v, 3) for v in per_rank_ms],
                       "allgather_ms": round(1000.0 * statistics.median(gathers_max), 3),
                       "allgather_bytes": int(gathered.numel() * 4), "rccl_direct": rccl_direct, "config3_e2e": e2e3,
                       "backend": (dist.get_backend() if dist else None), "rccl_ranks": (dist.get_world_size() if dist else 1),
                       "latents_sha256_16": hashlib.sha256(gathered.detach().cpu().contiguous().numpy().tobytes()).hexdigest()[:16],
                       "engine_build": builds, "preflight": preflight},
            "roofline": roof, "cpu_baseline": cpu, "parity_check": parity, "secondary": secondary,
        }
        print(json.dumps(line), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
