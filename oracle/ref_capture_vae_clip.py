"""Golden capture for the VAE decoder (G9) and CLIP text encoder (G10) by importing the reference.
Build container only; writes tests/golden/vae.npz and tests/golden/clip.npz.  See oracle/ref_capture.py."""
import json
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
    OUT = ref_capture.OUT
    # ---- G9: VAE decode through the reference's own VAE class (fp32 on CPU) -------------------------
    from src.AutoEncoders import VariationalAE
    g = {}
    for ch in (64, 128):
        cfg = ldx.VAEConfig(ch=ch)
        dd = {"double_z": True, "z_channels": 4, "resolution": 256, "in_channels": 3, "out_ch": 3, "ch": ch,
              "ch_mult": [1, 2, 4, 4], "num_res_blocks": 2, "attn_resolutions": [], "dropout": 0.0}
        eng = VariationalAE.AutoencodingEngine(VariationalAE.Encoder(**dd), VariationalAE.Decoder(**dd),
                                               VariationalAE.DiagonalGaussianRegularizer())
        sd = ldx.weights.synth_state_dict(ldx.weights.vae_decoder_state_dict_spec(cfg), seed=4321, dtype=torch.float32)
        missing, unexpected = eng.load_state_dict(sd, strict=False)
        assert not unexpected and all(k.startswith(("encoder.", "quant_conv.")) for k in missing), (missing[:5], unexpected[:5])
        eng = eng.eval().float()
        gen = torch.Generator().manual_seed(ch)
        z = torch.randn([1, 4, 8, 12], generator=gen)
        with torch.no_grad():
            y = torch.clamp((eng.decode(z) + 1.0) / 2.0, 0.0, 1.0).movedim(1, -1)          # VAE.decode post-process (:715-721)
        g[f"z_{ch}"] = z.numpy(); g[f"img_{ch}"] = y.numpy()
    np.savez_compressed(os.path.join(OUT, "vae.npz"), **g)
    print("vae.npz", {k: v.shape for k, v in g.items()})

    # ---- G10: tokenizer ids + CLIP conditioning through SD1ClipModel / SDTokenizer --------------------
    from src.SD15 import SDClip, SDToken
    tok = SDToken.SD1Tokenizer()
    prompts = ["a photo of a (red:1.3) cat, masterpiece", "", "plain prompt without weights",
               " ".join(["word%d" % i for i in range(120)]), "escaped \\(parens\\) and [brackets], embedding:missing"]
    g = {}
    pairs_all = []
    for i, pmt in enumerate(prompts):
        t = tok.tokenize_with_weights(pmt)["l"]
        pairs_all.append([[(int(a[0]), float(a[1])) for a in chunk] for chunk in t])
        g[f"ids_{i}"] = np.array([[a[0] for a in chunk] for chunk in t], dtype=np.int64)
        g[f"wts_{i}"] = np.array([[a[1] for a in chunk] for chunk in t], dtype=np.float32)
    g["prompts"] = np.array(prompts)
    for name, ccfg in (("tiny", ldx.CLIPConfig.tiny()),):
        cjson = {"num_hidden_layers": ccfg.num_layers, "hidden_size": ccfg.hidden_size, "num_attention_heads": ccfg.num_heads,
                 "intermediate_size": ccfg.intermediate_size, "hidden_act": "quick_gelu", "max_position_embeddings": 77,
                 "eos_token_id": 2}
        path = "/tmp/ldx_ref_scratch/clip_tiny.json"
        json.dump(cjson, open(path, "w"))
        model = SDClip.SD1ClipModel(device="cpu", dtype=torch.float16, textmodel_json_config=path)
        sd = ldx.weights.synth_state_dict(ldx.weights.clip_state_dict_spec(ccfg), seed=777)
        sd_ref = {(k if k.startswith("text_projection") else "text_model." + k): v for k, v in sd.items()}
        m, u = model.clip_l.transformer.load_state_dict(sd_ref, strict=True)
        for skip in (None, -2):
            model.reset_clip_options()
            if skip is not None:
                model.set_clip_options({"layer": skip})
            for i in range(len(prompts)):
                with torch.no_grad():
                    cond, pooled = model.encode_token_weights({"l": [[(a, b) for a, b in chunk] for chunk in pairs_all[i]]})
                g[f"cond_{name}_skip{skip}_{i}"] = cond.float().numpy()
                g[f"pooled_{name}_skip{skip}_{i}"] = pooled.float().numpy()
    np.savez_compressed(os.path.join(OUT, "clip.npz"), **g)
    print("clip.npz", {k: v.shape for k, v in g.items() if k.startswith(("ids", "cond_tiny_skip-2"))})


if __name__ == "__main__":
    main()
