"""20-step end-to-end goldens of BASELINE configs 1 and 2 from the REFERENCE, at full width (859.5 M parameters).

Test infrastructure only (build container only: imports /root/reference through oracle/ref_capture.enter_reference).
Writes tests/golden/unet_full20.npz — outputs, the two prompt tensors and the reference's wall time only:

  ks64_20_out    KSampler.sample (sampling.py:773-887 -> samplers.py:166-327), sample_euler / normal, 20 steps, cfg 7,
                 seed 42, multiscale off, zero latent 64^2   = BASELINE config 1 (SD1.5 512x512, 20 Euler steps, bs 1)
  ks128_20_out   the same at latent 128^2                     = BASELINE config 2, the headline (1024x1024, 20 steps, bs 1)
  ks*_trace      the denoised prediction's per-step rms (cheap side channel: which step a divergence starts at)
  P, N           positive / negative conditioning, randn Generator(7) as in ref_capture_full.py

Weights: ldx.weights.synth_state_dict(spec, seed=1234), fp16 storage, fp32 compute (manual_cast) — what bench.py times.
bench.py compares the latents it produces with exactly this configuration against ks128_20_out ("parity_check").

Usage:  python oracle/ref_capture_full20.py [--only64]        (64^2: ~2 min, 128^2: ~20 min on 8 cores)
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
OUT = os.path.join(REPO, "tests", "golden", "unet_full20.npz")


def prompts():
    """Shared with the tests and bench.py: a pure function (no reference import)."""
    gen = torch.Generator().manual_seed(7)
    P = torch.randn([1, 77, 768], generator=gen)
    N = torch.randn([1, 77, 768], generator=gen)
    return P, N


def main():
    import ldx_amd as ldx
    import ref_capture as RC
    torch.set_num_threads(8)
    RC.enter_reference()
    from src.sample import sampling
    cfg = ldx.UNetConfig.sd15()
    sd = ldx.weights.synth_state_dict(ldx.weights.unet_state_dict_spec(cfg), seed=1234)
    model, mp = RC.build_reference_model(cfg, sd)
    P, N = prompts()
    z = torch.zeros(1, 768)
    pos = [[P, {"pooled_output": z}]]
    neg = [[N, {"pooled_output": z}]]
    g = dict(np.load(OUT)) if os.path.exists(OUT) else {}
    timing = json.loads(str(g["timing_json"])) if "timing_json" in g else {}
    timing.update({"host": os.uname().nodename, "cpus": os.cpu_count(), "torch_threads": torch.get_num_threads(),
                   "torch": torch.__version__})
    g["P"], g["N"] = P.numpy(), N.numpy()

    # side channel: rms of what the model returns at every call of the hook position (cond.py:254-263 calls apply_model)
    trace = []
    orig = model.apply_model

    def traced(*a, **k):
        o = orig(*a, **k)
        trace.append(float(o.float().pow(2).mean().sqrt()))
        return o

    model.apply_model = traced
    for lat in ([64] if "--only64" in sys.argv else [64, 128]):
        trace.clear()
        t0 = time.perf_counter()
        with torch.no_grad():
            o = sampling.KSampler().sample(model=mp, seed=42, steps=20, cfg=7.0, denoise=1.0, positive=pos, negative=neg,
                                           latent_image={"samples": torch.zeros(1, 4, lat, lat)}, pipeline=True,
                                           disable_pbar=True, sampler_name="sample_euler", scheduler="normal",
                                           enable_multiscale=False)
        dt = (time.perf_counter() - t0) / 20
        assert len(trace) == 20, len(trace)
        g[f"ks{lat}_20_out"] = o[0]["samples"].numpy()
        g[f"ks{lat}_20_trace"] = np.array(trace, dtype=np.float32)
        timing[f"ks{lat}_20_s_per_step"] = round(dt, 2)
        print(f"ks{lat}_20 done, {dt:.1f} s/step, |out| max {float(o[0]['samples'].abs().max()):.3f}", flush=True)
        g["timing_json"] = np.array(json.dumps(timing))
        np.savez_compressed(OUT, **g)
    print("wrote", OUT, {k: getattr(v, "shape", None) for k, v in g.items()}, timing)


if __name__ == "__main__":
    main()
