"""c_concat branch of BaseModel.apply_model (src/Model/ModelBase.py:100-101) — oracle vs the reference's own outputs (tests/golden/concat.npz, captured
by oracle/ref_capture_concat.py on a tiny 9-channel UNet).  CPU only; fp32 vs fp32: rtol 1e-4 like the other one-forward goldens."""
import dataclasses
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import sd15_oracle as O  # noqa: E402


@pytest.mark.parametrize("tag", ["a", "b"])
def test_oracle_apply_model_with_c_concat(ldx, golden_dir, tag):
    g = np.load(os.path.join(golden_dir, "concat.npz"))
    cfg = dataclasses.replace(ldx.UNetConfig.tiny(64, 128), in_channels=9)
    sd = ldx.weights.synth_state_dict(ldx.weights.unet_state_dict_spec(cfg), seed=4321)
    t = lambda k: torch.from_numpy(g[f"{tag}_{k}"])
    with torch.no_grad():
        out = O.apply_model(sd, cfg, t("x"), t("sigma"), t("ctx"), c_concat=t("cc"))
    rel = float((out.double() - t("out").double()).norm() / t("out").double().norm())
    assert rel <= 1e-4, rel
    # the concat channels are NOT scaled by 1 / sqrt(sigma^2 + 1): scaling them moves the result well beyond the parity error
    s = t("sigma").view(-1, 1, 1, 1)
    with torch.no_grad():
        wrong = O.apply_model(sd, cfg, t("x"), t("sigma"), t("ctx"), c_concat=t("cc") / (s ** 2 + 1.0) ** 0.5)
    wrel = float((wrong.double() - t("out").double()).norm() / t("out").double().norm())
    assert wrel > 2e-4 and wrel > 20 * rel, (wrel, rel)
