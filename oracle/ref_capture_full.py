"""Full-width (model_channels = 320, 859.5 M parameters) goldens from the REFERENCE, for the benchmarked network.

Test infrastructure only (build container only: imports /root/reference through oracle/ref_capture.enter_reference).
Writes tests/golden/unet_full.npz with seeds + outputs only (inputs are regenerated from the seeds by the tests):

  am64_*   model.apply_model (ModelBase.py:72-133 -> unet.py:679-770) at latent 64^2  (BASELINE config 1 shape), CFG batch 2
  am128_*  the same at latent 128^2 (BASELINE config 2 = the headline shape), CFG batch 2
  ks64_*   KSampler.sample, sample_euler / normal, 4 steps (steps 20 of the schedule truncated would change sigmas;
           a plain 4-step schedule is used), cfg 7, latent 64^2, seed 42  (a3-a8 at full width)

Weights: ldx.weights.synth_state_dict(spec, seed=1234) in fp16 storage, fp32 compute (manual_cast) — the same
generator bench.py uses.  Also records the reference's own wall time per evaluation on this container's cores
(the `reference_cpu` figure bench.py reports next to the port's).

Usage:  python oracle/ref_capture_full.py [--skip128]
"""
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))
OUT = os.path.join(REPO, "tests", "golden")


def inputs(lat, seed, ctx_tokens=77):
    """Shared with tests/test_fullwidth_gpu.py: inputs are a pure function of (lat, seed)."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn([2, 4, lat, lat], generator=g)
    ctx = torch.randn([2, ctx_tokens, 768], generator=g)
    return x, ctx


def main():
    import ldx_amd as ldx
    import ref_capture as RC
    torch.set_num_threads(8)
    RC.enter_reference()
    from src.sample import sampling
    cfg = ldx.UNetConfig.sd15()
    sd = ldx.weights.synth_state_dict(ldx.weights.unet_state_dict_spec(cfg), seed=1234)
    model, mp = RC.build_reference_model(cfg, sd)
    g = {}
    timing = {"host": os.uname().nodename, "cpus": os.cpu_count(), "torch_threads": torch.get_num_threads(),
              "torch": torch.__version__}

    cases = [(64, 101, [7.5, 7.5]), (64, 102, [0.3, 0.3])]
    if "--skip128" not in sys.argv:
        cases.append((128, 103, [5.0, 5.0]))
    for lat, seed, sig in cases:
        x, ctx = inputs(lat, seed)
        sigma = torch.tensor(sig)
        with torch.no_grad():
            t0 = time.perf_counter()
            out = model.apply_model(x, sigma, c_crossattn=ctx, transformer_options={})
            dt = time.perf_counter() - t0
        key = f"am{lat}_{seed}"
        g[key + "_sigma"] = sigma.numpy()
        g[key + "_out"] = out.float().numpy()
        timing[key + "_s"] = round(dt, 2)
        print(key, "done in", round(dt, 1), "s", float(out.abs().max()), flush=True)

    # a3-a8 at full width: 4-step sample_euler / normal at 64^2
    gen = torch.Generator().manual_seed(7)
    P = torch.randn([1, 77, 768], generator=gen)
    N = torch.randn([1, 77, 768], generator=gen)
    z = torch.zeros(1, 768)
    pos = [[P, {"pooled_output": z}]]
    neg = [[N, {"pooled_output": z}]]
    t0 = time.perf_counter()
    with torch.no_grad():
        o = sampling.KSampler().sample(model=mp, seed=42, steps=4, cfg=7.0, denoise=1.0, positive=pos, negative=neg,
                                       latent_image={"samples": torch.zeros(1, 4, 64, 64)}, pipeline=True,
                                       disable_pbar=True, sampler_name="sample_euler", scheduler="normal",
                                       enable_multiscale=False)
    timing["ks64_s_per_step"] = round((time.perf_counter() - t0) / 4, 2)
    g["ks64_P"] = P.numpy()
    g["ks64_N"] = N.numpy()
    g["ks64_out"] = o[0]["samples"].numpy()
    print("ks64 done", timing["ks64_s_per_step"], "s/step", flush=True)

    g["timing_json"] = np.array(json.dumps(timing))
    np.savez_compressed(os.path.join(OUT, "unet_full.npz"), **g)
    print("wrote unet_full.npz", {k: getattr(v, "shape", None) for k, v in g.items()}, timing)


if __name__ == "__main__":
    main()
