"""bench.py — sampler it/s for SD1.5 1024x1024 bs=1 (BASELINE.json metric) on N MI355X GPUs of one node.

One "step" = one sampler-loop iteration as the reference's tqdm counts it (samplers.py:269): one
CFG-batched UNet evaluation ([uncond; cond] -> batch 2, cond.py:186-226) through ldx_unet_denoise plus the
fused CFG + Euler update (ldx_sampler_step).  Full-size SD1.5 layout (859.5 M params), synthetic seeded
weights and inputs (no checkpoints offline), latent [1,4,128,128], 77-token contexts, cfg 7, sample_euler /
normal schedule, multiscale off = SURVEY.md §8(d) config 2.  Inputs are resident in HBM before timing.

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL), each rank denoises its own latent
(batch shard, bs = N total, weak scaling, no per-step collective); one all-gather of the final latents
closes the timed region (north_star: "all-gather of decoded latents only").

Prints ONE JSON line on rank 0 (contract in the task prompt) with two extra objects:
  roofline     — dominant kernel = the single (kernel, op shape) with the largest share of step time (the level-0
                 self-attention, D = 40, 16384 tokens), algorithmic FLOP/s from per-op HIP events recorded on the launch
                 stream (ldx_profile mode 2), against the dense bf16 MFMA peak (2.5 PFLOP/s, MI355X_MICROARCH.md); HBM
                 traffic per launch from the committed PMC passes (profiles/r*/traffic.json); the per-class breakdown in
                 `kernels`; plus whole-step achieved TFLOP/s (9.348 TFLOP/step).
  cpu_baseline — the oracle (CPU restatement, kind "port") timed on this host's cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0     # dense MFMA bf16, MI355X_MICROARCH.md "Chip-level parameters"


def cpu_baseline(cfg, sd, budget_s=25.0):
    """Oracle apply_model (CFG batch 2) on the host cores.  Tries 512^2 first, then the full 1024^2 workload
    if one 512^2 evaluation took less than a quarter of the budget."""
    from oracle import sd15_oracle as O      # baseline leg only — never on the product path
    threads = torch.get_num_threads()
    g = torch.Generator().manual_seed(7)
    ctx = torch.randn([2, 77, cfg.context_dim], generator=g)
    sdf = {k: v.float() for k, v in sd.items()}          # cast once (reference re-casts per call; excluded here)
    result = None
    for lat, label in ((64, "512x512"), (128, "1024x1024")):
        x = torch.randn([2, 4, lat, lat], generator=g)
        sig = torch.tensor([5.0, 5.0])
        t0 = time.perf_counter()
        with torch.no_grad():
            O.apply_model(sdf, cfg, x, sig, ctx)
        dt = time.perf_counter() - t0
        result = {"value": round(1.0 / dt, 5), "unit": "it/s", "cores": threads, "kind": "port",
                  "sample": f"1 CFG-batched UNet evaluation (oracle.apply_model, fp32) at {label}, batch 2, "
                            f"{dt:.2f} s on {threads} torch threads of {os.cpu_count()} host cpus"}
        if dt > budget_s / 4:
            break
    return result


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--latent", type=int, default=128)
    ap.add_argument("--batch", type=int, default=1, help="images per GPU (default 1 = the headline config; 8 = the per-GPU shard of SURVEY config 3)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--tiny", action="store_true", help="debug: tiny UNet config")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist_mod.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        dist = dist_mod
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import ldx_amd as ldx
    cfg = ldx.UNetConfig.tiny(64, 128) if args.tiny else ldx.UNetConfig.sd15()
    spec = ldx.weights.unet_state_dict_spec(cfg)
    sd = ldx.weights.synth_state_dict(spec, seed=1234)
    eng = ldx.UNetEngine(cfg, sd, device=local_rank, dtype=args.dtype)
    if not args.no_graph:
        eng.set_graph_mode(True)

    lat = args.latent
    total = args.warmup + args.steps
    ms = ldx.sampling.ModelSamplingDiscrete()
    sigmas = ldx.sampling.calculate_sigmas(ms, "normal", total)
    g = torch.Generator().manual_seed(7)
    pos = torch.randn([1, 77, cfg.context_dim], generator=g)
    neg = torch.randn([1, 77, cfg.context_dim], generator=g)
    pb = args.batch
    noise = ldx.parallel.shard_noise((world * pb, 4, lat, lat), 42, rank, world)              # config-3 style shard
    x = (noise * torch.sqrt(1.0 + sigmas[0] ** 2.0)).to(dev)
    model = ldx.sampling.CFGDenoiser(eng, pos, neg, 7.0, pb, lat, lat)

    def run_steps(i0, n):
        for i in range(i0, i0 + n):
            du, dc = model(x, sigmas[i])
            ldx.sampling._step(0, x, du, dc, 7.0, sigmas[i], sigmas[i + 1] - sigmas[i])

    run_steps(0, args.warmup)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    run_steps(args.warmup, args.steps)
    gathered = ldx.parallel.gather_latents(x, world, dist)    # final latents only: the job's single collective
    assert gathered.shape[0] == world * pb
    ev1.record()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert torch.isfinite(x).all(), "non-finite latents"

    info = eng.plan_info()
    # ---- per-kernel roofline leg: same process, same data, HIP events per op on the launch stream ----
    roof = None
    if rank == 0:
        eng.set_graph_mode(False)
        eng._lib.ldx_profile(eng._h, 2, 1)            # mode 2: keyed by kernel class AND op shape
        xs = x.clone()
        nprof = 3
        for i in range(nprof):
            du, dc = model(xs, sigmas[args.warmup])
        torch.cuda.synchronize()
        eng.profile(False, reset=False)
        rep = eng.profile_report()
        tot_ms = sum(v["ms"] for v in rep.values())
        # per kernel class (shape suffix stripped) for the breakdown ...
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
        # ... and the dominant KERNEL = the single (kernel, shape) with the most time: one launch geometry, so that "per launch"
        # flops, duration and PMC traffic all refer to the same thing (the level-0 self-attention at 1024^2)
        dom = max((k for k in rep if rep[k]["flops"] > 0), key=lambda k: rep[k]["ms"])
        dv = rep[dom]
        dom_tflops = dv["flops"] / (dv["ms"] * 1e-3) / 1e12
        traffic = None
        try:
            import glob
            tj = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "traffic.json")))[-1]
            traffic = json.load(open(tj)).get(dom, {}).get("bytes")      # HBM bytes per launch from the committed PMC passes
        except Exception:
            traffic = None
        step_ms_gpu = ev0.elapsed_time(ev1) / args.steps
        roof = {"bound": "mfma", "kernel": dom, "achieved": round(dom_tflops, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                "frac": round(dom_tflops / PEAK_BF16_TFLOPS, 4), "traffic": traffic,
                "launches_per_step": dv["count"] // nprof, "avg_launch_ms": round(dv["ms"] / dv["count"], 4),
                "flop_per_launch": round(dv["flops"] / dv["count"] / 1e9, 2),
                "share_of_step": round(dv["ms"] / tot_ms, 4),
                "step_tflop": round(info["flops"] / 1e12, 4),
                "step_achieved": round(info["flops"] / (step_ms_gpu * 1e-3) / 1e12, 2),
                "step_frac": round(info["flops"] / (step_ms_gpu * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
                "kernels": kern}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(cfg, sd)

    if rank == 0:
        value = world * args.steps / elapsed          # sampler iterations per second (each iteration advances `batch` images per GPU)
        line = {
            "metric": "sampler it/s (UNet steps/sec) SD1.5 1024x1024 bs=1 bf16",
            "value": round(value, 3), "unit": "it/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1000.0 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": (round(value / 2.8, 3) if world == 1 else None), "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"SD1.5 UNet (859.5M params, synthetic seeded weights) sampler loop, latent "
                                   f"[{pb},4,{lat},{lat}] ({lat * 8}x{lat * 8}), CFG batch {2 * pb}, ctx 77x768, sample_euler/normal, "
                                   f"multiscale off; per-GPU bs={pb}, {world * pb} image(s) in flight",
                       "global_batch": world * pb, "images_per_gpu": pb, "image_steps_per_s": round(world * pb * args.steps / elapsed, 3),
                       "parallelism": f"batch-shard x{world} (replicated weights, final all-gather)",
                       "launches_per_step": info["launches"], "hip_graph": not args.no_graph,
                       "vs_baseline_note": "2.8 it/s = README table, RTX 3060 mobile + Stable-Fast (BASELINE.md §1)"},
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
