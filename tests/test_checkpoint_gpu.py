"""Checkpoint ingestion end to end on the GPU: a full SD1.5-layout state dict (model.diffusion_model.* /
first_stage_model.* / cond_stage_model.transformer.*) on disk -> split -> layout sniffing -> LoRA merge (the golden LoRA of
tests/golden/lora.npz, whose merged tensors are pinned to the reference's calculate_weight) -> engines, compared with the
oracle run on the same merged weights.  No real checkpoint exists offline: the file holds seeded synthetic weights."""
import os

import numpy as np
import pytest
import torch

from oracle import sd15_oracle as O

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def test_safetensors_checkpoint_with_lora_through_the_engines(ldx, ldx_lib, golden_dir, tmp_path):
    import safetensors.torch
    ck = ldx.checkpoint
    ucfg, vcfg, ccfg = ldx.UNetConfig.tiny(64, 128), ldx.VAEConfig(ch=64), ldx.CLIPConfig.tiny()
    usd = ldx.weights.synth_state_dict(ldx.weights.unet_state_dict_spec(ucfg), seed=1234)
    vsd = ldx.weights.synth_state_dict(ldx.weights.vae_decoder_state_dict_spec(vcfg), seed=4321, dtype=torch.float32)
    csd = ldx.weights.synth_state_dict(ldx.weights.clip_state_dict_spec(ccfg), seed=777)
    full = {ck.UNET_PREFIX + k: v for k, v in usd.items()}
    full.update({ck.VAE_PREFIX + k: v for k, v in vsd.items()})
    full.update({ck.CLIP_PREFIX + "text_model." + k: v for k, v in csd.items()})
    path = os.path.join(str(tmp_path), "synthetic_sd15.safetensors")
    safetensors.torch.save_file({k: v.contiguous() for k, v in full.items()}, path)

    unet, vae, clip = ck.split_sd15_checkpoint(safetensors.torch.load_file(path, device="cpu"))
    assert ck.detect_unet_config(unet) == ucfg
    g = np.load(os.path.join(golden_dir, "lora.npz"))
    lora = {k[len("lora::"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("lora::")}
    merged, n = ck.merge_lora({k: v.float() for k, v in unet.items()}, lora, ck.lora_key_map_unet(ucfg, unet.keys()), strength=0.7)
    assert n == 4
    for i in range(4):          # the merged tensors are the reference's (ModelPatcher.calculate_weight)
        assert torch.allclose(merged[str(g[f"target_{i}"])], torch.from_numpy(g[f"merged_{i}"]), rtol=1e-6, atol=1e-7)
    merged16 = {k: v.half() for k, v in merged.items()}         # what ModelPatcher.patch_model leaves in the fp16 module

    gen = torch.Generator().manual_seed(5)
    x = torch.randn([2, 4, 16, 16], generator=gen); sigma = torch.tensor([2.0, 2.0]); ctx = torch.randn([2, 77, 128], generator=gen)
    with torch.no_grad():
        ref = O.apply_model(merged16, ucfg, x, sigma, ctx)
        base = O.apply_model(usd, ucfg, x, sigma, ctx)
    assert _rel(base, ref) > 1e-4                               # the LoRA changes the output: the merge is actually exercised
    for dt, tol in (("f16", 4e-3), ("bf16", 2.5e-2)):
        eng = ldx.UNetEngine(ucfg, merged16, device=0, dtype=dt)
        out = eng.denoise(x.cuda(), sigma.cuda(), ctx.cuda())
        r = _rel(out, ref)
        print(f"[{dt}] checkpoint + LoRA -> UNetEngine vs oracle on the merged weights: rel-L2 {r:.3e}")
        assert r <= tol
    # the other two parts of the file reach their engines unchanged
    z = torch.randn([1, 4, 8, 8], generator=gen)
    img = ldx.VAEDecoderEngine(vcfg, vae, device=0, dtype="f16").decode(z.cuda())
    with torch.no_grad():
        assert _rel(img, O.vae_decode(vae, vcfg, z)) <= 1e-2
    ids = torch.tensor([[49406, 320, 1125, 49407] + [49407] * 73])
    last, _, _ = ldx.CLIPTextEngine(ccfg, clip, device=0, dtype="f16").forward(ids)
    with torch.no_grad():
        want, _, _ = O.clip_forward({k[len("text_model."):] if k.startswith("text_model.") else k: v for k, v in clip.items()}, ccfg, ids)
    assert _rel(last, want) <= 4e-3
