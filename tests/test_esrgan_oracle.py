"""Pin the ESRGAN restatements (oracle.rrdbnet_forward, oracle.tiled_scale) and the host-side key renaming against the
reference's RRDBNet / tiled_scale (tests/golden/esrgan.npz from oracle/ref_capture_esrgan.py).  CPU only."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import sd15_oracle as O  # noqa: E402


def _rel(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.fixture(scope="module")
def setup(ldx, golden_dir):
    cfg = ldx.ESRGANConfig.tiny()
    sd = ldx.weights.synth_state_dict(ldx.weights.esrgan_state_dict_spec(cfg), seed=77, dtype=torch.float32)
    return cfg, sd, np.load(os.path.join(golden_dir, "esrgan.npz"))


def test_rrdbnet_forward(setup):
    cfg, sd, g = setup
    with torch.no_grad():
        y = O.rrdbnet_forward(sd, cfg, torch.from_numpy(g["x"]))
    assert y.shape == g["y"].shape and _rel(y, g["y"]) < 1e-5


def test_tiled_scale(setup):
    cfg, sd, g = setup
    fn = lambda a: O.rrdbnet_forward(sd, cfg, a)      # noqa: E731
    with torch.no_grad():
        s = O.tiled_scale(torch.from_numpy(g["img"]), fn, tile_x=32, tile_y=32, overlap=8, upscale_amount=4)
        s1 = O.tiled_scale(torch.from_numpy(g["img"])[:, :, :30, :28], fn, tile_x=32, tile_y=32, overlap=8, upscale_amount=4)
    assert _rel(s, g["tiled"]) < 1e-5 and _rel(s1, g["single"]) < 1e-5


def test_new_arch_key_renaming(setup, ldx):
    """weights.esrgan_new_to_old_arch == RRDBNet.new_to_old_arch on Real-ESRGAN style names (the only ones the reference reads)."""
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
    assert sorted(new.keys()) == [str(k) for k in g["new_keys"]]
    old = ldx.weights.esrgan_new_to_old_arch(new)
    assert sorted(old.keys()) == sorted(sd.keys()) and all(torch.equal(old[k], sd[k]) for k in sd)


def test_x4plus_layout(ldx):
    spec = ldx.weights.esrgan_state_dict_spec(ldx.ESRGANConfig())
    assert ldx.weights.param_count(spec) == 16_697_987            # RealESRGAN_x4plus / ESRGAN x4 (23 RRDB blocks)
