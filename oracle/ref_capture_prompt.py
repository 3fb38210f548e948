"""Golden capture for the prompt-weight parser and CLIP chunking by importing the reference's
SDTokenizer (src/SD15/SDToken.py) with its CLIPTokenizerFast vocabulary.  Stores, per prompt, the (token, weight) chunks
and the parsed (segment, weight) list, plus the per-word token ids of every word that occurs (so the tests need no
vocabulary files).  Build container only; writes tests/golden/prompt.npz.  See oracle/ref_capture.py."""
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "oracle"))
import ref_capture  # noqa: E402

PROMPTS = [
    "a photo of a (red:1.3) cat, masterpiece",
    "",
    "plain prompt without weights",
    "((very)) detailed (oil painting:0.8) of a [castle] on a hill, (dramatic (storm:1.5) clouds), 8k",
    "escaped \\(parentheses\\) stay (literal \\(here\\):1.2)",
    "unbalanced (open group never closes, and a stray ) too",
    "(bad:weight) (half:.5) (neg:-1) (trailing colon:) (:1.4)",
    "line one\nline two (with\nnewline:1.1)   multiple   spaces",
    " ".join(["word%d" % i for i in range(90)]),
    "supercalifragilisticexpialidocious " * 12 + "(antidisestablishmentarianism pneumonoultramicroscopicsilicovolcanoconiosis:1.7)",
    "embedding:missing_one, then text",
]


def main():
    ref_capture.enter_reference()
    from src.SD15 import SDToken
    tok = SDToken.SDTokenizer()
    vocab, g = {}, {}
    enc = lambda w: tok.tokenizer(w)["input_ids"][tok.tokens_start:-1]      # noqa: E731
    for i, p in enumerate(PROMPTS):
        chunks = tok.tokenize_with_weights(p)
        g[f"ids_{i}"] = np.array([[t for t, _ in c] for c in chunks], dtype=np.int64)
        g[f"wts_{i}"] = np.array([[w for _, w in c] for c in chunks], dtype=np.float64)
        parsed = SDToken.token_weights(SDToken.escape_important(p), 1.0)
        g[f"parsed_{i}"] = np.array(json.dumps([[SDToken.unescape_important(s), w] for s, w in parsed]))
        for seg, _ in parsed:
            for w in SDToken.unescape_important(seg).replace("\n", " ").split(" "):
                if w and w not in vocab:
                    vocab[w] = [int(t) for t in enc(w)]
    # with an embedding directory that does not hold the name, the word is dropped (SDToken.py:322-334) and the comma survives
    import tempfile
    tok2 = SDToken.SDTokenizer(embedding_directory=tempfile.mkdtemp())
    chunks = tok2.tokenize_with_weights(PROMPTS[-1])
    g["ids_missing"] = np.array([[t for t, _ in c] for c in chunks], dtype=np.int64)
    vocab.setdefault(",", [int(t) for t in enc(",")])
    g["prompts"] = np.array(PROMPTS)
    g["vocab"] = np.array(json.dumps(vocab))
    g["start_end"] = np.array([tok.start_token, tok.end_token])
    np.savez_compressed(os.path.join(ref_capture.OUT, "prompt.npz"), **g)
    print("prompt.npz", len(vocab), "words;", [g[f"ids_{i}"].shape for i in range(len(PROMPTS))])


if __name__ == "__main__":
    main()
