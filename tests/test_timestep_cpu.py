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


def test_host_timestep_index_on_scheduler_sigmas(ldx):
    """Every sigma a sampler loop can hand over: normal / karras / simple schedules for 1..28 steps (sampling.calculate_sigmas, itself pinned to the
    reference's tables in test_sampling_host.py).  engine.timestep_index == the reference's expression with a correctly rounded logarithm, restated in
    numpy (host-independent), and == sampling.ModelSamplingDiscrete.timestep (torch's fp32 log: the reference's literal expression) wherever that log is
    correctly rounded on this host — at most a near-tie may differ, and then only by one index."""
    ms = ldx.sampling.ModelSamplingDiscrete()
    ls = ms.log_sigmas.numpy()
    n = differ = 0
    for name in ("normal", "karras", "simple"):
        for steps in (1, 2, 3, 8, 20, 28):
            sig = ldx.sampling.calculate_sigmas(ms, name, steps)
            sig = sig[sig > 0]
            got = ldx.engine.timestep_index(ms.log_sigmas, sig)
            lg = np.log(sig.numpy().astype(np.float64)).astype(np.float32)
            want = np.abs(lg[None, :] - ls[:, None]).argmin(axis=0)
            assert np.array_equal(got.numpy(), want), (name, steps)
            lit = ms.timestep(sig)
            d = (got - lit).abs()
            assert int(d.max()) <= 1, (name, steps)
            differ += int((d != 0).sum()); n += sig.numel()
    assert differ <= max(1, n // 100), f"{differ} of {n} scheduler sigmas differ from torch's fp32-log expression"
