"""ESRGAN RRDBNet (ldx_esrgan_forward) and the feathered tiled_scale (ldx_tile_blend / ldx_tile_finish) on a real MI355X vs
the reference goldens.  Tolerances: activations are 16 bit through 2 RRDB blocks (30 convs): rel-L2 <= 4e-3 (fp16) / 2.5e-2 (bf16)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import sd15_oracle as O  # noqa: E402  (checker only)


def _rel(a, b):
    a, b = a.double().cpu(), torch.as_tensor(b).double()
    return float((a - b).norm() / b.norm())


@pytest.fixture(scope="module")
def setup(ldx, ldx_lib, golden_dir):
    cfg = ldx.ESRGANConfig.tiny()
    sd = ldx.weights.synth_state_dict(ldx.weights.esrgan_state_dict_spec(cfg), seed=77, dtype=torch.float32)
    return cfg, sd, np.load(os.path.join(golden_dir, "esrgan.npz"))


@pytest.mark.parametrize("dt,tol", [("f16", 4e-3), ("bf16", 2.5e-2)])
def test_rrdbnet_and_tiled_scale_vs_reference(ldx, setup, dt, tol):
    cfg, sd, g = setup
    eng = ldx.ESRGANEngine(cfg, sd, device=0, dtype=dt)
    x = torch.from_numpy(g["x"]).movedim(1, -1).contiguous().cuda()              # NHWC at the ABI
    y = eng.forward(x).movedim(-1, 1)
    r = _rel(y, g["y"])
    # tiled_scale incl. ImageUpscaleWithModel's final clamp (USDU_upscaler.py:94)
    img = torch.from_numpy(g["img"]).movedim(1, -1).contiguous()
    s = eng.upscale(img, tile=32, overlap=8).movedim(-1, 1)
    rt = _rel(s, np.clip(g["tiled"], 0.0, 1.0))
    s1 = eng.upscale(img[:, :30, :28, :], tile=32, overlap=8).movedim(-1, 1)
    r1 = _rel(s1, np.clip(g["single"], 0.0, 1.0))
    print(f"[{dt}] RRDBNet rel-L2 {r:.3e}; tiled_scale {rt:.3e}; single tile {r1:.3e}")
    assert y.shape == g["y"].shape and r <= tol and rt <= tol and r1 <= tol
    assert float(s.min()) >= 0.0 and float(s.max()) <= 1.0


def test_new_arch_names_and_odd_sizes(ldx, setup):
    """Real-ESRGAN key names load through the host renaming; odd image sizes vs the oracle."""
    cfg, sd, g = setup
    new = {}
    for k, v in sd.items():
        p = k.split(".")
        if k.startswith("model.0."): new["conv_first." + p[-1]] = v
        elif k.startswith(f"model.1.sub.{cfg.num_blocks}."): new["conv_body." + p[-1]] = v
        elif k.startswith("model.1.sub."): new[f"body.{p[3]}.rdb{p[4][3]}.conv{p[5][4]}.{p[-1]}"] = v
        elif k.startswith("model.3."): new["conv_up1." + p[-1]] = v
        elif k.startswith("model.6."): new["conv_up2." + p[-1]] = v
        elif k.startswith("model.8."): new["conv_hr." + p[-1]] = v
        elif k.startswith("model.10."): new["conv_last." + p[-1]] = v
    eng = ldx.ESRGANEngine(cfg, new, device=0, dtype="f16")
    x = torch.rand(1, 3, 17, 23, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        ref = O.rrdbnet_forward(sd, cfg, x)
    y = eng.forward(x.movedim(1, -1).contiguous().cuda()).movedim(-1, 1)
    assert _rel(y, ref) <= 4e-3
