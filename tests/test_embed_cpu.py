"""Textual-inversion embeddings on the CPU: the file loader, the "embedding:name" words of the tokenizer and the
oracle's token-table extension against reference-captured goldens (oracle/ref_capture_embed.py, which wrote the same files
through the same helper and ran the reference's load_embed / SDTokenizer / SD1ClipModel on them)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import sd15_oracle as O


def _rel(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def write_embedding_files(d, E, t):
    """One file per branch of load_embed (same layouts as oracle/ref_capture_embed.py::write_embedding_files)."""
    import safetensors.torch
    safetensors.torch.save_file({"emb_params": t["styleA"]}, os.path.join(d, "styleA.safetensors"))
    torch.save({"string_to_param": {"*": t["styleB"]}, "name": "styleB"}, os.path.join(d, "styleB.pt"))
    torch.save([{"clip_l": t["styleC"][:1], "clip_g": torch.zeros(1, E + 8)}, {"clip_l": t["styleC"][1:]}], os.path.join(d, "styleC.pt"))
    os.makedirs(os.path.join(d, "sub"), exist_ok=True)
    torch.save({"clip_g": torch.zeros(2, E), "clip_l": t["styleD"]}, os.path.join(d, "sub", "styleD.bin"))
    safetensors.torch.save_file({"emb_params": t["wrongsize"]}, os.path.join(d, "wrongsize.safetensors"))


@pytest.fixture(scope="module")
def emb(golden_dir, tmp_path_factory):
    g = np.load(os.path.join(golden_dir, "embed.npz"))
    E = int(g["E"])
    d = str(tmp_path_factory.mktemp("embeddings"))
    write_embedding_files(d, E, {k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("t_")})
    return g, E, d


def test_load_embed_formats(ldx, emb):
    g, E, d = emb
    for name, key in (("styleA", None), ("styleB", "clip_l"), ("styleC.pt", "clip_l"), ("styleD", "clip_l"), ("wrongsize", None), ("nope", None),
                      ("../styleA", None)):
        want = g["load_" + name.replace("/", "_").replace(".", "_")]
        got = ldx.checkpoint.load_embed(name, d, E, key)
        if want.size == 0:
            assert got is None, name
        else:
            assert got is not None and tuple(got.shape) == want.shape and np.array_equal(got.float().numpy(), want), name
    assert ldx.checkpoint.load_embed("styleA", [d], E) is not None           # a list of directories works like a string
    with open(os.path.join(d, "broken.pt"), "wb") as f:
        f.write(b"not a checkpoint")
    assert ldx.checkpoint.load_embed("broken", d, E) is None                 # unreadable file: skipped, like the reference


def _chunks(ldx, g, i, E, d):
    vocab = json.loads(str(g["vocab"]))
    return ldx.prompt.tokenize_with_weights(str(g["prompts"][i]), lambda w: vocab[w],
                                            embeddings=ldx.checkpoint.EmbeddingDirectory(d, embedding_size=E))


def test_tokenizer_embedding_words(ldx, emb):
    g, E, d = emb
    for i in range(len(g["prompts"])):
        chunks = _chunks(ldx, g, i, E, d)
        vecs = [np.array(v, dtype=np.float32) for v in json.loads(str(g[f"vec_{i}"]))]
        ids, wts = g[f"ids_{i}"], g[f"wts_{i}"]
        assert len(chunks) == ids.shape[0]
        for c, chunk in enumerate(chunks):
            assert len(chunk) == ids.shape[1]
            for j, (t, w) in enumerate(chunk):
                assert w == wts[c][j]
                if ids[c][j] < 0:                                            # a vector token
                    assert not isinstance(t, int) and np.array_equal(torch.as_tensor(t).float().numpy(), vecs[-ids[c][j] - 1])
                else:
                    assert int(t) == ids[c][j]


def test_clip_with_embeddings_oracle(ldx, emb):
    g, E, d = emb
    cfg = ldx.CLIPConfig.tiny()
    sd = ldx.weights.synth_state_dict(ldx.weights.clip_state_dict_spec(cfg), seed=777)
    for i in range(len(g["prompts"])):
        with torch.no_grad():
            cond, pooled = O.clip_encode_token_weights(sd, cfg, _chunks(ldx, g, i, E, d), layer_idx=-2)
        assert cond.shape == g[f"cond_{i}"].shape
        assert _rel(cond, g[f"cond_{i}"]) < 1e-5 and _rel(pooled, g[f"pooled_{i}"]) < 1e-5
