"""Golden capture for the ESRGAN upscaler: the reference's RRDBNet built from a synthetic state dict (2 RRDB
blocks, x4) and its tiled_scale with feathered blending.  Build container only; writes tests/golden/esrgan.npz."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "oracle"))
import ref_capture  # noqa: E402


def main():
    sys.path.insert(0, REPO)
    import ldx_amd as ldx
    torch.set_num_threads(8)
    ref_capture.enter_reference()
    from src.UltimateSDUpscale import RDRB
    from src.Utilities import util
    cfg = ldx.ESRGANConfig.tiny()
    sd = ldx.weights.synth_state_dict(ldx.weights.esrgan_state_dict_spec(cfg), seed=77, dtype=torch.float32)
    # the reference only ingests the Real-ESRGAN / BSRGAN ("new arch") key names and renames them to its module names
    # (new_to_old_arch, RDRB.py:355-441); feed it the synthetic weights under those names
    new = {}
    for k, v in sd.items():
        parts = k.split(".")
        if k.startswith("model.0."): new["conv_first." + parts[-1]] = v
        elif k.startswith(f"model.1.sub.{cfg.num_blocks}."): new["conv_body." + parts[-1]] = v
        elif k.startswith("model.1.sub."): new[f"body.{parts[3]}.rdb{parts[4][3]}.conv{parts[5][4]}.{parts[-1]}"] = v
        elif k.startswith("model.3."): new["conv_up1." + parts[-1]] = v
        elif k.startswith("model.6."): new["conv_up2." + parts[-1]] = v
        elif k.startswith("model.8."): new["conv_hr." + parts[-1]] = v
        elif k.startswith("model.10."): new["conv_last." + parts[-1]] = v
    model = RDRB.RRDBNet(dict(new)).eval()
    assert model.scale == 4 and model.num_blocks == cfg.num_blocks
    assert sorted(model.state.keys()) == sorted(sd.keys())          # the renamed keys are exactly the engine's key set
    g = {}
    gen = torch.Generator().manual_seed(4)
    x = torch.rand([2, 3, 24, 20], generator=gen)
    with torch.no_grad():
        y = model(x)
    g["x"] = x.numpy(); g["y"] = y.numpy()
    g["new_keys"] = np.array(sorted(new.keys()))
    img = torch.rand([1, 3, 40, 56], generator=gen)
    with torch.no_grad():
        s = util.tiled_scale(img, lambda a: model(a), tile_x=32, tile_y=32, overlap=8, upscale_amount=4)
        s1 = util.tiled_scale(img[:, :, :30, :28], lambda a: model(a), tile_x=32, tile_y=32, overlap=8, upscale_amount=4)   # single-tile shortcut
    g["img"] = img.numpy(); g["tiled"] = s.numpy(); g["single"] = s1.numpy()
    np.savez_compressed(os.path.join(ref_capture.OUT, "esrgan.npz"), **g)
    print("esrgan.npz", {k: v.shape for k, v in g.items()})


if __name__ == "__main__":
    main()
