"""Prompt-weight syntax and CLIP chunking against the reference's SDTokenizer output
(tests/golden/prompt.npz from oracle/ref_capture_prompt.py).  Token ids and weights must match exactly; the per-word
BPE ids come from the fixture's vocabulary, so no tokenizer files are needed.  CPU only."""
import json
import os

import numpy as np
import pytest


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "prompt.npz"))


def test_parse_prompt_weights_matches_reference(ldx, g):
    from ldx_amd import prompt
    for i, p in enumerate(g["prompts"]):
        want = [(s, w) for s, w in json.loads(str(g[f"parsed_{i}"]))]
        got = prompt.parse_prompt_weights(str(p))
        assert [s for s, _ in got] == [s for s, _ in want], p
        assert [w for _, w in got] == [w for _, w in want], p      # same float operations: exactly equal


def test_tokenize_with_weights_matches_reference(ldx, g):
    from ldx_amd import prompt
    vocab = json.loads(str(g["vocab"]))
    start, end = (int(v) for v in g["start_end"])
    for i, p in enumerate(g["prompts"]):
        chunks = prompt.tokenize_with_weights(str(p), lambda w: vocab[w], start_token=start, end_token=end)      # no embedding directory
        ids = np.array([[t for t, _ in c] for c in chunks], dtype=np.int64)
        wts = np.array([[w for _, w in c] for c in chunks], dtype=np.float64)
        assert ids.shape == g[f"ids_{i}"].shape, (p, ids.shape)
        assert np.array_equal(ids, g[f"ids_{i}"]), p
        assert np.array_equal(wts, g[f"wts_{i}"]), p


def test_unknown_embedding_is_dropped_like_the_reference(ldx, g):
    from ldx_amd import prompt
    vocab = json.loads(str(g["vocab"]))
    start, end = (int(v) for v in g["start_end"])
    chunks = prompt.tokenize_with_weights(str(g["prompts"][-1]), lambda w: vocab[w], start_token=start, end_token=end, embeddings={})
    assert np.array_equal(np.array([[t for t, _ in c] for c in chunks], dtype=np.int64), g["ids_missing"])


def test_known_answers(ldx):
    from ldx_amd import prompt
    assert prompt.parse_prompt_weights("a (b:1.5) c") == [("a ", 1.0), ("b", 1.5), (" c", 1.0)]
    assert prompt.parse_prompt_weights("((x))") == [("x", 1.1 * 1.1)]
    assert prompt.parse_prompt_weights("\\(x\\)") == [("(x)", 1.0)]
    enc = lambda w: [100 + len(w)] * len(w)      # noqa: E731  toy tokenizer: one token per character
    one = prompt.tokenize_with_weights("ab (cd:2)", enc)
    assert len(one) == 1 and len(one[0]) == 77
    assert one[0][:6] == [(49406, 1.0), (102, 1.0), (102, 1.0), (102, 2.0), (102, 2.0), (49407, 1.0)] and one[0][-1] == (49407, 1.0)
    # a 7-token word that does not fit is moved whole to the next chunk; an 8-token one is split
    filler = " ".join(["abcde"] * 14)            # 70 tokens
    moved = prompt.tokenize_with_weights(filler + " abcdefg", enc)
    assert len(moved) == 2 and moved[0][71] == (49407, 1.0) and moved[1][1:8] == [(107, 1.0)] * 7
    split = prompt.tokenize_with_weights(filler + " abcdefgh", enc)
    assert len(split) == 2 and split[0][71:76] == [(108, 1.0)] * 5 and split[0][76] == (49407, 1.0) and split[1][1:4] == [(108, 1.0)] * 3
    # textual-inversion rows enter as tokens of their own; an unknown name is dropped
    import torch
    emb = {"style": torch.ones(2, 4)}
    with_emb = prompt.tokenize_with_weights("x embedding:style, (embedding:nope:1.2) y", enc, embeddings=emb)
    toks = [t for t, _ in with_emb[0]]
    assert sum(isinstance(t, torch.Tensor) for t in toks) == 2 and toks[4] == 101      # "," kept as leftover text
