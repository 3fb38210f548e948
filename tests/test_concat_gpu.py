"""c_concat through the HIP engine and the hook (ldx_unet_denoise_concat; reference: BaseModel.apply_model src/Model/ModelBase.py:100-101, reached
through the wrapper call of src/cond/cond.py:254-263 with c = {"c_concat": ..., "c_crossattn": ...}) against the reference's own outputs
(tests/golden/concat.npz).  Tolerances as the other one-forward goldens: fp16-activation mode rel-L2 <= 4e-3, bf16 <= 2.5e-2."""
import dataclasses
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.double().cpu(), torch.as_tensor(b).double()
    return float((a - b).norm() / b.norm())


@pytest.fixture(scope="module")
def setup(ldx, ldx_lib, golden_dir):
    cfg = dataclasses.replace(ldx.UNetConfig.tiny(64, 128), in_channels=9)
    sd = ldx.weights.synth_state_dict(ldx.weights.unet_state_dict_spec(cfg), seed=4321)
    return np.load(os.path.join(golden_dir, "concat.npz")), {dt: ldx.UNetEngine(cfg, sd, device=0, dtype=dt) for dt in ("f16", "bf16")}


@pytest.mark.parametrize("dt,tol", [("f16", 4e-3), ("bf16", 2.5e-2)])
@pytest.mark.parametrize("tag", ["a", "b"])
def test_engine_and_hook_with_c_concat(ldx, setup, dt, tol, tag):
    g, eng = setup
    t = lambda k: torch.from_numpy(g[f"{tag}_{k}"])
    e = eng[dt]
    out = e.denoise(t("x").cuda(), t("sigma").cuda(), t("ctx").cuda(), c_concat=t("cc").cuda())
    r = _rel(out, g[f"{tag}_out"])
    # the hook, called the way cond.py:254-263 calls it, with CPU tensors like a CPU-device run of the reference
    patch = ldx.LdxUNetPatch(e)
    hooked = patch(None, {"input": t("x"), "timestep": t("sigma"), "c": {"c_crossattn": t("ctx"), "c_concat": t("cc")}, "cond_or_uncond": [0] * t("x").shape[0]})
    assert hooked.device.type == "cpu" and torch.equal(hooked, out.cpu())
    print(f"[{dt}] c_concat {tag}: rel-L2 {r:.3e}")
    assert r <= tol, r
    # graph replay binds the c_concat pointer too
    e.set_graph_mode(True)
    try:
        xs, ss, cs, ccs = t("x").cuda(), t("sigma").cuda(), t("ctx").cuda(), t("cc").cuda()
        o2 = torch.empty_like(out)
        for _ in range(3):
            e.denoise(xs, ss, cs, out=o2, c_concat=ccs)
        assert torch.equal(o2, out)
        other = (ccs * 0.5).contiguous()
        o3 = e.denoise(xs, ss, cs, c_concat=other)
        assert not torch.equal(o3, out)
    finally:
        e.set_graph_mode(False)


def test_concat_argument_checks(ldx, setup):
    g, eng = setup
    e = eng["f16"]
    x, s, c = torch.from_numpy(g["a_x"]).cuda(), torch.from_numpy(g["a_sigma"]).cuda(), torch.from_numpy(g["a_ctx"]).cuda()
    with pytest.raises(AssertionError):
        e.denoise(x, s, c, c_concat=torch.zeros(2, 3, 16, 16).cuda())           # 4 + 3 != in_channels
    with pytest.raises(AssertionError):
        e.denoise(x, s, c)                                                      # a 9-channel engine without c_concat: x would need 9 channels
