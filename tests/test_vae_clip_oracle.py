"""Pin the VAE-decode and CLIP oracle restatements against reference-captured goldens (CPU only).
Goldens: oracle/ref_capture_vae_clip.py (imports /root/reference in the build container)."""
import os

import numpy as np
import pytest
import torch

from oracle import sd15_oracle as O


def _rel(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.parametrize("ch", [64, 128])
def test_vae_decode_oracle(ldx, golden_dir, ch):
    g = np.load(os.path.join(golden_dir, "vae.npz"))
    cfg = ldx.VAEConfig(ch=ch)
    sd = ldx.weights.synth_state_dict(ldx.weights.vae_decoder_state_dict_spec(cfg), seed=4321, dtype=torch.float32)
    with torch.no_grad():
        img = O.vae_decode(sd, cfg, torch.from_numpy(g[f"z_{ch}"]))
    assert img.shape == g[f"img_{ch}"].shape and img.min() >= 0 and img.max() <= 1
    assert _rel(img, g[f"img_{ch}"]) < 1e-4


def test_tokenizer_kat(golden_dir):
    """KAT recorded in SURVEY.md Appendix B for "a photo of a (red:1.3) cat, masterpiece"."""
    g = np.load(os.path.join(golden_dir, "clip.npz"))
    ids, wts = g["ids_0"][0], g["wts_0"][0]
    assert list(ids[:10]) == [49406, 320, 1125, 539, 320, 736, 2368, 267, 12066, 49407] and all(ids[10:] == 49407)
    assert abs(wts[5] - 1.3) < 1e-6 and wts[4] == 1.0
    assert g["ids_3"].shape == (5, 77)         # 120 words -> several 77-token chunks


@pytest.mark.parametrize("skip", [None, -2])
def test_clip_conditioning_oracle(ldx, golden_dir, skip):
    g = np.load(os.path.join(golden_dir, "clip.npz"))
    cfg = ldx.CLIPConfig.tiny()
    sd = ldx.weights.synth_state_dict(ldx.weights.clip_state_dict_spec(cfg), seed=777)
    for i in range(5):
        pairs = [list(zip(g[f"ids_{i}"][c].tolist(), g[f"wts_{i}"][c].tolist())) for c in range(g[f"ids_{i}"].shape[0])]
        with torch.no_grad():
            cond, pooled = O.clip_encode_token_weights(sd, cfg, pairs, layer_idx=skip)
        want = g[f"cond_tiny_skip{skip}_{i}"]
        assert cond.shape == want.shape
        assert _rel(cond, want) < 1e-5
        assert _rel(pooled, g[f"pooled_tiny_skip{skip}_{i}"]) < 1e-5
