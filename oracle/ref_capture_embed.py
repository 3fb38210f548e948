"""Golden capture for textual-inversion embeddings by importing the reference: load_embed's file-format
rules (src/SD15/SDToken.py:125-206), the "embedding:name" words of SDTokenizer.tokenize_with_weights (:296-391) and the
token-table extension of SDClipModel.set_up_textual_embeddings (src/SD15/SDClip.py:213-267) as seen through
SD1ClipModel.encode_token_weights.  Build container only; writes tests/golden/embed.npz (the embedding tensors, the
token / weight chunks with vector tokens marked, the expected conditioning).  See oracle/ref_capture.py."""
import json
import os
import sys
import tempfile

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "oracle"))
import ref_capture  # noqa: E402

PROMPTS = [
    "a photo of embedding:styleA cat",
    "(embedding:styleB:1.3) and embedding:styleA, trailing comma and embedding:nope",
    "embedding:wrongsize next to (embedding:styleC.pt:0.7) " + " ".join("w%d" % i for i in range(70)),
]


def write_embedding_files(d, E, tensors):
    """The same files tests/test_embed_cpu.py writes: one per branch of load_embed."""
    import safetensors.torch
    safetensors.torch.save_file({"emb_params": tensors["styleA"]}, os.path.join(d, "styleA.safetensors"))      # first value of a plain dict
    torch.save({"string_to_param": {"*": tensors["styleB"]}, "name": "styleB"}, os.path.join(d, "styleB.pt"))  # A1111 layout
    torch.save([{"clip_l": tensors["styleC"][:1], "clip_g": torch.zeros(1, E + 8)}, {"clip_l": tensors["styleC"][1:]}],
               os.path.join(d, "styleC.pt"))                                                                   # list of dicts: rows of width E are kept
    os.makedirs(os.path.join(d, "sub"), exist_ok=True)
    torch.save({"clip_g": torch.zeros(2, E), "clip_l": tensors["styleD"]}, os.path.join(d, "sub", "styleD.bin"))  # embed_key picks clip_l; sub-directory
    safetensors.torch.save_file({"emb_params": tensors["wrongsize"]}, os.path.join(d, "wrongsize.safetensors"))


def main():
    sys.path.insert(0, REPO)
    import ldx_amd as ldx
    ref_capture.enter_reference()
    from src.SD15 import SDClip, SDToken
    ccfg = ldx.CLIPConfig.tiny()
    E = ccfg.hidden_size
    gen = torch.Generator().manual_seed(99)
    tensors = {"styleA": torch.randn(2, E, generator=gen) * 0.05, "styleB": torch.randn(3, E, generator=gen) * 0.05,
               "styleC": torch.randn(2, E, generator=gen) * 0.05, "styleD": torch.randn(1, E, generator=gen) * 0.05,
               "wrongsize": torch.randn(2, E + 16, generator=gen) * 0.05}
    d = tempfile.mkdtemp()
    write_embedding_files(d, E, tensors)
    g = {"E": np.array(E)}
    for k, v in tensors.items():
        g["t_" + k] = v.numpy()
    # ---- load_embed: what the reference returns per file layout ----
    for name, key in (("styleA", None), ("styleB", "clip_l"), ("styleC.pt", "clip_l"), ("styleD", "clip_l"), ("wrongsize", None), ("nope", None),
                      ("../styleA", None)):
        out = SDToken.load_embed(name, d, E, key)
        g["load_" + name.replace("/", "_").replace(".", "_")] = np.zeros((0, 0), np.float32) if out is None else out.float().numpy()
    # ---- tokenizer + CLIP with embedding words ----
    tok = SDToken.SDTokenizer(embedding_directory=d, embedding_size=E)
    cjson = {"num_hidden_layers": ccfg.num_layers, "hidden_size": E, "num_attention_heads": ccfg.num_heads,
             "intermediate_size": ccfg.intermediate_size, "hidden_act": "quick_gelu", "max_position_embeddings": 77, "eos_token_id": 2}
    path = "/tmp/ldx_ref_scratch/clip_tiny.json"
    json.dump(cjson, open(path, "w"))
    model = SDClip.SD1ClipModel(device="cpu", dtype=torch.float16, textmodel_json_config=path)
    sd = ldx.weights.synth_state_dict(ldx.weights.clip_state_dict_spec(ccfg), seed=777)
    model.clip_l.transformer.load_state_dict({(k if k.startswith("text_projection") else "text_model." + k): v for k, v in sd.items()}, strict=True)
    model.set_clip_options({"layer": -2})
    vocab = {}
    for i, p in enumerate(PROMPTS):
        chunks = tok.tokenize_with_weights(p)
        # ids: integer tokens as they are; a vector token is stored as -(1 + index into vec_<i>)
        vecs, ids, wts = [], [], []
        for c in chunks:
            row = []
            for t, w in c:
                if isinstance(t, torch.Tensor):
                    vecs.append(t.float().numpy()); row.append(-len(vecs))
                else:
                    row.append(int(t))
            ids.append(row); wts.append([float(w) for _, w in c])
        g[f"ids_{i}"] = np.array(ids, dtype=np.int64); g[f"wts_{i}"] = np.array(wts, dtype=np.float64)
        g[f"vec_{i}"] = np.array(json.dumps([v.tolist() for v in vecs]))            # ragged (the wrong-size rows differ in width)
        with torch.no_grad():
            cond, pooled = model.encode_token_weights({"l": chunks})
        g[f"cond_{i}"] = cond.float().numpy(); g[f"pooled_{i}"] = pooled.float().numpy()
        for seg, _ in SDToken.token_weights(SDToken.escape_important(p), 1.0):
            for w in SDToken.unescape_important(seg).replace("\n", " ").split(" "):
                if w and not w.startswith("embedding:") and w not in vocab:
                    vocab[w] = [int(t) for t in tok.tokenizer(w)["input_ids"][tok.tokens_start:-1]]
    vocab.setdefault(",", [int(t) for t in tok.tokenizer(",")["input_ids"][tok.tokens_start:-1]])
    g["prompts"] = np.array(PROMPTS); g["vocab"] = np.array(json.dumps(vocab))
    np.savez_compressed(os.path.join(ref_capture.OUT, "embed.npz"), **g)
    print("embed.npz", {k: v.shape for k, v in g.items() if k.startswith(("ids", "cond", "load"))})


if __name__ == "__main__":
    main()
