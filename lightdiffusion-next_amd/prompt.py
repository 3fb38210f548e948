"""Prompt-weight syntax and CLIP chunking on the host: "(text)", "((text))", "(text:1.3)", "\\(", "\\)" and the
77-token batching of SDTokenizer.tokenize_with_weights (src/SD15/SDToken.py:13-104, 296-391).  The BPE vocabulary itself is
not part of this package: `encode_word(word) -> [token ids]` is supplied by the caller (the reference passes
CLIPTokenizerFast(word)["input_ids"][1:-1]); everything around it — segmentation, weights, word-keeping across chunk
borders, start / end / pad tokens — is restated here and pinned against the reference's output (tests/golden/prompt.npz).
The result feeds CLIPTextEngine.encode_token_weights."""
from typing import Callable, Dict, List, Optional, Sequence, Tuple

_ESC_CLOSE, _ESC_OPEN = "\0\1", "\0\2"          # placeholders for "\)" and "\(" while parsing (SDToken.py:79-104)


def _split_top_level(text: str) -> List[str]:
    """Top-level pieces of `text`: runs outside parentheses and whole "( ... )" groups (nesting kept inside the group).
    Mirrors the reference's depth counter exactly, including what it does with unbalanced input: a group that never closes
    is emitted as it stands, a stray ")" is kept in the running piece and drives the depth negative."""
    pieces, cur, depth = [], "", 0
    for ch in text:
        if ch == "(":
            if depth == 0:
                if cur:
                    pieces.append(cur)
                cur = "("
            else:
                cur += ch
            depth += 1
        elif ch == ")":
            depth -= 1
            if depth == 0:
                pieces.append(cur + ")")
                cur = ""
            else:
                cur += ch
        else:
            cur += ch
    if cur:
        pieces.append(cur)
    return pieces


def _weights(text: str, weight: float) -> List[Tuple[str, float]]:
    out: List[Tuple[str, float]] = []
    for piece in _split_top_level(text):
        if len(piece) >= 2 and piece[0] == "(" and piece[-1] == ")":
            inner, w = piece[1:-1], weight * 1.1
            colon = inner.rfind(":")
            if colon > 0:
                try:
                    w = float(inner[colon + 1:])
                    inner = inner[:colon]
                except ValueError:
                    pass
            out += _weights(inner, w)
        else:
            out.append((piece, weight))
    return out


def parse_prompt_weights(text: str) -> List[Tuple[str, float]]:
    """[(segment, weight)] as token_weights(escape_important(text), 1.0) gives them, segments un-escaped."""
    esc = text.replace("\\)", _ESC_CLOSE).replace("\\(", _ESC_OPEN)
    return [(seg.replace(_ESC_CLOSE, ")").replace(_ESC_OPEN, "("), w) for seg, w in _weights(esc, 1.0)]


def tokenize_with_weights(text: str, encode_word: Callable[[str], Sequence[int]], start_token: Optional[int] = 49406,
                          end_token: int = 49407, max_length: int = 77, pad_with_end: bool = True, pad_to_max_length: bool = True,
                          min_length: Optional[int] = None, max_word_length: int = 8,
                          embeddings: Optional[Dict[str, Sequence]] = None) -> List[List[Tuple[object, float]]]:
    """SDTokenizer.tokenize_with_weights (SDToken.py:296-391): chunks of `max_length` (token, weight) pairs.
    Words of fewer than `max_word_length` tokens are never split across chunks (the chunk is closed and padded instead);
    longer ones are.  `embeddings`: textual-inversion vectors by name for "embedding:name" words (each row becomes one
    token whose id is the vector itself, as in the reference); an unknown name is dropped, like the reference does."""
    pad_token = end_token if pad_with_end else 0
    words: List[List[Tuple[object, float]]] = []
    for segment, weight in parse_prompt_weights(text):
        for word in segment.replace("\n", " ").split(" "):
            if word == "":
                continue
            if word.startswith("embedding:") and embeddings is not None:
                name = word[len("embedding:"):].strip("\n")
                emb, leftover = embeddings.get(name), ""
                if emb is None and name.strip(",") != name:       # "name," -> try "name", keep the commas as text
                    stripped = name.strip(",")
                    emb, leftover = embeddings.get(stripped), name[len(stripped):]
                if emb is not None:
                    rows = [emb] if getattr(emb, "ndim", 2) == 1 else [emb[i] for i in range(len(emb))]
                    words.append([(r, weight) for r in rows])
                if leftover == "":
                    continue
                word = leftover
            words.append([(t, weight) for t in encode_word(word)])

    def new_chunk():
        return [(start_token, 1.0)] if start_token is not None else []

    chunks = [new_chunk()]
    cur = chunks[0]
    for group in words:
        keep_whole = len(group) < max_word_length
        while group:
            if len(group) + len(cur) > max_length - 1:
                room = max_length - len(cur) - 1
                if not keep_whole:
                    cur.extend(group[:room])
                    cur.append((end_token, 1.0))
                    group = group[room:]
                else:
                    cur.append((end_token, 1.0))
                    if pad_to_max_length:
                        cur.extend([(pad_token, 1.0)] * room)
                cur = new_chunk()
                chunks.append(cur)
            else:
                cur.extend(group)
                group = []
    cur.append((end_token, 1.0))
    if pad_to_max_length:
        cur.extend([(pad_token, 1.0)] * (max_length - len(cur)))
    if min_length is not None and len(cur) < min_length:
        cur.extend([(pad_token, 1.0)] * (min_length - len(cur)))
    return chunks
