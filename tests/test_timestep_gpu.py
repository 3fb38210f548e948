"""The device side of the sigma -> timestep index lookup: integer work, compared as integers.

ldx_unet_timestep runs the SAME device function the boundary kernel of ldx_unet_denoise runs (csrc/misc.hip nearest_log_sigma); here it sees every golden
sigma of tests/golden/schedules.npz (28 table points, 24 random sigmas, 8 geometric midpoints = near-ties) and every sigma of the normal / karras / simple
schedules for 1, 2, 3, 8, 20, 28 steps, and must agree with the reference's indices (golden / the oracle) exactly.  Where sigma is a HOST value the engine
does not rely on this at all: denoise_cfg / denoise(host sigma) pass the index computed with the reference's own torch expression
(ldx_unet_denoise_cfg_t / ldx_unet_denoise_t); the last test checks that path end to end against ldx_unet_forward at the golden index."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import sd15_oracle as O  # noqa: E402  (checker only)


@pytest.fixture(scope="module")
def eng(ldx, ldx_lib):
    cfg = ldx.UNetConfig.tiny(64, 128)
    sd = ldx.weights.synth_state_dict(ldx.weights.unet_state_dict_spec(cfg), seed=1234)
    return cfg, ldx.UNetEngine(cfg, sd, device=0, dtype="f16")


def test_device_lookup_equals_reference_on_all_golden_sigmas(eng, golden_dir):
    cfg, e = eng
    sched = np.load(os.path.join(golden_dir, "schedules.npz"))
    sig = torch.from_numpy(sched["timestep_in"].copy())
    got = e.timestep_device(sig).cpu().numpy()
    bad = np.nonzero(got != sched["timestep_out"])[0]
    print(f"device lookup on {sig.numel()} golden sigmas: {bad.size} mismatches {[(float(sig[i]), int(got[i]), int(sched['timestep_out'][i])) for i in bad[:8]]}")
    assert np.array_equal(got, sched["timestep_out"].astype(got.dtype))


def test_device_lookup_equals_oracle_on_every_scheduler_sigma(eng, ldx):
    cfg, e = eng
    ms = ldx.sampling.ModelSamplingDiscrete()
    n = 0
    for name in ("normal", "karras", "simple"):
        for steps in (1, 2, 3, 8, 20, 28):
            sig = ldx.sampling.calculate_sigmas(ms, name, steps)
            sig = sig[sig > 0].contiguous()
            want = O.timestep(sig).numpy()
            got = e.timestep_device(sig).cpu().numpy()
            assert np.array_equal(got, want.astype(got.dtype)), (name, steps, sig.tolist(), got.tolist(), want.tolist())
            n += sig.numel()
    print(f"device lookup == oracle on {n} scheduler sigmas")


def test_denoise_paths_use_the_reference_index(eng, golden_dir):
    """denoise_cfg(sigma: float) and denoise(sigma on the host) at ALL 60 golden sigmas are bit-identical to ldx_unet_denoise_t with the GOLDEN index
    (same kernels, same inputs: the index is the only thing that can differ), and the index really reaches the boundary kernel (index + 1 changes the output)."""
    cfg, e = eng
    sched = np.load(os.path.join(golden_dir, "schedules.npz"))
    sig_all = torch.from_numpy(sched["timestep_in"].copy())
    tt_all = torch.from_numpy(sched["timestep_out"].copy())
    gen = torch.Generator().manual_seed(2)
    x = torch.randn([1, 4, 8, 8], generator=gen).cuda()
    xx = torch.cat([x, x]).contiguous()
    ctx = torch.randn([2, 77, cfg.context_dim], generator=gen).cuda()
    e.set_cfg_share(False)            # compare like with like: ldx_unet_denoise_t runs every op on the full batch
    try:
        for i in range(sig_all.numel()):
            s = float(sig_all[i])
            sg = torch.full((2,), s)
            want = e._run(e._lib.ldx_unet_denoise, xx, sg, ctx, None, t_idx=tt_all[i].repeat(2).float()).clone()
            assert torch.equal(e.denoise_cfg(x, s, ctx), want), (i, s)
            assert torch.equal(e.denoise(xx, sg, ctx), want), (i, s)                      # host sigma -> host index -> ldx_unet_denoise_t
            if i % 10 == 0:
                other = e._run(e._lib.ldx_unet_denoise, xx, sg, ctx, None, t_idx=(tt_all[i].repeat(2).float() + 1).clamp(max=999))
                assert not torch.equal(other, want) or int(tt_all[i]) == 999
    finally:
        e.set_cfg_share(True)
