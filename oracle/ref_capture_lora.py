"""Golden capture for checkpoint ingestion: the reference's LoRA key maps, load_lora and
ModelPatcher.calculate_weight on the tiny UNet with a synthetic LoRA.  Build container only; writes tests/golden/lora.npz."""
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
    ref_capture.enter_reference()
    from src.Model import LoRas
    cfg = ldx.UNetConfig.tiny(64, 128)
    sd = ldx.weights.synth_state_dict(ldx.weights.unet_state_dict_spec(cfg), seed=1234)
    model, mp = ref_capture.build_reference_model(cfg, sd)
    key_map = LoRas.model_lora_keys_unet(mp.model, {})
    g = {"map_keys": np.array(sorted(key_map.keys())), "map_vals": np.array([key_map[k] for k in sorted(key_map.keys())])}
    gen = torch.Generator().manual_seed(8)
    # a synthetic LoRA: kohya (diffusers-style) names for two attention projections and a conv, ldm-style for one, one without alpha
    targets = {
        "lora_unet_down_blocks_0_attentions_0_transformer_blocks_0_attn1_to_q": ("input_blocks.1.1.transformer_blocks.0.attn1.to_q.weight", 4, 2.0),
        "lora_unet_mid_block_attentions_0_transformer_blocks_0_attn2_to_k": ("middle_block.1.transformer_blocks.0.attn2.to_k.weight", 8, 8.0),
        "lora_unet_up_blocks_1_resnets_0_conv1": ("output_blocks.3.0.in_layers.2.weight", 4, 1.0),
        "lora_unet_input_blocks_4_1_proj_in": ("input_blocks.4.1.proj_in.weight", 2, None),
    }
    lora = {}
    for name, (tkey, rank, alpha) in targets.items():
        w = sd[tkey]
        out_f, in_f = w.shape[0], int(np.prod(w.shape[1:]))
        lora[f"{name}.lora_up.weight"] = torch.randn(out_f, rank, generator=gen) * 0.1
        lora[f"{name}.lora_down.weight"] = torch.randn(rank, in_f, generator=gen) * 0.1
        if alpha is not None:
            lora[f"{name}.alpha"] = torch.tensor(alpha)
    loaded = LoRas.load_lora(lora, key_map)
    mpc = mp.clone()
    mpc.add_patches(loaded, 0.7)
    for i, (name, (tkey, rank, alpha)) in enumerate(targets.items()):
        full = "diffusion_model." + tkey
        w = model.state_dict()[full].clone()
        merged = mpc.calculate_weight(mpc.patches[full], w.float(), full)
        g[f"merged_{i}"] = merged.numpy(); g[f"target_{i}"] = np.array(tkey)
    for k, v in lora.items():
        g["lora::" + k] = v.numpy()
    np.savez_compressed(os.path.join(ref_capture.OUT, "lora.npz"), **g)
    print("lora.npz", len(key_map), "map entries;", [str(g[f"target_{i}"]) for i in range(4)])


if __name__ == "__main__":
    main()
