"""Checkpoint ingestion: LoRA key maps and merge vs the reference's model_lora_keys_unet / load_lora /
calculate_weight (tests/golden/lora.npz from oracle/ref_capture_lora.py), checkpoint splitting and layout sniffing.  CPU only."""
import os

import numpy as np
import pytest
import torch


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "lora.npz"))


@pytest.fixture(scope="module")
def tiny(ldx):
    cfg = ldx.UNetConfig.tiny(64, 128)
    return cfg, ldx.weights.synth_state_dict(ldx.weights.unet_state_dict_spec(cfg), seed=1234)


def test_lora_key_map_equals_reference(ldx, g, tiny):
    cfg, sd = tiny
    ck = ldx.checkpoint
    mine = ck.lora_key_map_unet(cfg, sd.keys())
    ref = {str(k): str(v)[len("diffusion_model."):] for k, v in zip(g["map_keys"], g["map_vals"])}
    ref = {k: v for k, v in ref.items() if not k.startswith("lora_prior_unet_")}          # cascade alias of the same entries
    # the reference also lists diffusers names whose target does not exist in this model (label_emb, absent skip convs ...);
    # they can never patch anything (ModelPatcher.add_patches skips unknown keys), so the engine-side map leaves them out
    assert all(v not in sd for k, v in ref.items() if k not in mine)
    assert mine == {k: v for k, v in ref.items() if v in sd}


def test_merge_lora_equals_reference(ldx, g, tiny):
    cfg, sd = tiny
    ck = ldx.checkpoint
    lora = {k[len("lora::"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("lora::")}
    merged, n = ck.merge_lora({k: v.float() for k, v in sd.items()}, lora, ck.lora_key_map_unet(cfg, sd.keys()), strength=0.7)
    assert n == 4
    for i in range(4):
        t = str(g[f"target_{i}"])
        assert torch.allclose(merged[t], torch.from_numpy(g[f"merged_{i}"]), rtol=1e-6, atol=1e-7), t
    untouched = [k for k in sd if k not in {str(g[f"target_{i}"]) for i in range(4)}]
    assert all(torch.equal(merged[k], sd[k].float()) for k in untouched[:50])


def test_split_and_detect(ldx, tiny):
    ck = ldx.checkpoint
    for cfg in (ldx.UNetConfig.tiny(64, 128), ldx.UNetConfig.sd15()):
        spec = ldx.weights.unet_state_dict_spec(cfg)
        full = {"model.diffusion_model." + k: torch.empty(s, dtype=torch.float16, device="meta") for k, s in spec}
        full["first_stage_model.decoder.conv_in.weight"] = torch.empty(1, device="meta")
        full["cond_stage_model.transformer.text_model.final_layer_norm.weight"] = torch.empty(1, device="meta")
        unet, vae, clip = ck.split_sd15_checkpoint(full)
        assert set(unet) == {k for k, _ in spec} and list(vae) == ["decoder.conv_in.weight"] and list(clip) == ["text_model.final_layer_norm.weight"]
        assert ck.detect_unet_config(unet) == cfg
