"""sigma -> timestep index is INTEGER work (ModelSamplingDiscrete.timestep, sampling.py:309-320): the host expression the
engine uses wherever sigma is a host value (engine.timestep_index -> ldx_unet_denoise_cfg_t / ldx_unet_denoise_t) must reproduce the reference's
indices bit for bit on every golden sigma — table points, random sigmas and the geometric midpoints (near-ties)."""
import os

import numpy as np
import torch


def test_host_timestep_index_equals_reference_on_all_golden_sigmas(ldx, golden_dir):
    sched = np.load(os.path.join(golden_dir, "schedules.npz"))
    sigmas, log_sigmas = ldx.engine.sd15_sigmas()
    got = ldx.engine.timestep_index(log_sigmas, torch.from_numpy(sched["timestep_in"])).numpy()
    assert got.dtype == np.int64 and np.array_equal(got, sched["timestep_out"])
    # one value at a time (how denoise_cfg calls it: a python float) gives the same indices
    one = np.array([int(ldx.engine.timestep_index(log_sigmas, float(s))[0]) for s in sched["timestep_in"]])
    assert np.array_equal(one, sched["timestep_out"])


def test_host_timestep_index_on_scheduler_sigmas_equals_sampling_module(ldx):
    """Every sigma a sampler loop can hand over: normal / karras / simple schedules for 1..28 steps (sampling.calculate_sigmas, itself pinned to the
    reference's tables in test_sampling_host.py) — engine.timestep_index == sampling.ModelSamplingDiscrete.timestep (the reference's expression)."""
    ms = ldx.sampling.ModelSamplingDiscrete()
    for name in ("normal", "karras", "simple"):
        for steps in (1, 2, 3, 8, 20, 28):
            sig = ldx.sampling.calculate_sigmas(ms, name, steps)
            sig = sig[sig > 0]
            want = ms.timestep(sig)
            got = ldx.engine.timestep_index(ms.log_sigmas, sig)
            assert torch.equal(got, want), (name, steps)
