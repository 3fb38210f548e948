"""GGUF container + Q8_0 blocks -> dense state dict (lightdiffusion-next_amd/gguf_loader.py; reference: Quantize/Quantizer.py:94-112, 581-665).

PARITY UNPINNED (the `gguf` package the reference's loader needs is not installed, so no golden can be captured from it): these tests pin the loader
against a file WRITTEN HERE from the published GGUF v3 layout and against the block arithmetic restated independently — d (fp16) -> dtype, q int8,
weight = d * q in that dtype, shapes = reversed GGUF dims unless `comfy.gguf.orig_shape.<name>` says otherwise, `model.diffusion_model.` stripped,
architecture checked against the reference's list."""
import struct

import numpy as np
import pytest
import torch


def _s(b: bytes) -> bytes:
    return struct.pack("<Q", len(b)) + b


def _q8_0(x: np.ndarray):
    """llama.cpp's Q8_0 quantiser on the last axis: per 32 values d = amax / 127 (stored fp16), q = round(x / d)."""
    xb = x.reshape(-1, 32).astype(np.float32)
    d = (np.abs(xb).max(axis=1) / 127.0).astype(np.float16)
    inv = np.where(d.astype(np.float32) > 0, 1.0 / np.maximum(d.astype(np.float32), 1e-30), 0.0)
    q = np.clip(np.rint(xb * inv[:, None]), -127, 127).astype(np.int8)
    raw = np.concatenate([d.view(np.uint8).reshape(-1, 2), q.view(np.uint8)], axis=1).reshape(-1)
    return raw, d, q


def _write(path, tensors, meta, version=3, align=32):
    """tensors: [(name, torch-order shape, ggml type, raw bytes)]"""
    out = bytearray(struct.pack("<IIQQ", 0x46554747, version, len(tensors), len(meta)))
    for k, (t, v) in meta.items():
        out += _s(k.encode()) + struct.pack("<I", t)
        if t == 8:
            out += _s(v.encode())
        elif t == 4:
            out += struct.pack("<I", v)
        elif t == 9:                                   # array of int32
            out += struct.pack("<IQ", 5, len(v)) + b"".join(struct.pack("<i", e) for e in v)
    off, blobs = 0, []
    for name, shape, typ, raw in tensors:
        dims = tuple(reversed(shape))                  # GGUF: innermost first
        out += _s(name.encode()) + struct.pack("<I", len(dims)) + b"".join(struct.pack("<Q", d) for d in dims) + struct.pack("<IQ", typ, off)
        pad = (-len(raw)) % align
        blobs.append(bytes(raw) + b"\0" * pad)
        off += len(raw) + pad
    out += b"\0" * ((-len(out)) % align)
    for b in blobs:
        out += b
    with open(path, "wb") as f:
        f.write(out)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32])
def test_q8_0_file_round_trip(ldx, tmp_path, dtype):
    G = ldx.gguf_loader
    rng = np.random.default_rng(3)
    w = rng.standard_normal((48, 64)).astype(np.float32) * 0.05          # a Linear weight [out][in], in % 32 == 0
    raw, d, q = _q8_0(w)
    bias = rng.standard_normal(48).astype(np.float32)
    norm = rng.standard_normal(64).astype(np.float16)
    conv = rng.standard_normal((8, 4, 3, 3)).astype(np.float16)
    path = str(tmp_path / "m.gguf")
    _write(path, [("model.diffusion_model.blk.weight", (48, 64), G.GGML_Q8_0, raw.tobytes()),
                  ("model.diffusion_model.blk.bias", (48,), G.GGML_F32, bias.tobytes()),
                  ("model.diffusion_model.norm.scale", (64,), G.GGML_F16, norm.tobytes()),
                  ("model.diffusion_model.conv.weight", (72, 4), G.GGML_F16, conv.tobytes()),      # stored flattened: the orig_shape metadata restores it
                  ("first_stage_model.other", (4,), G.GGML_F32, np.zeros(4, np.float32).tobytes())],
           {"general.architecture": (8, "flux"), "general.alignment": (4, 32),
            "comfy.gguf.orig_shape.model.diffusion_model.conv.weight": (9, [8, 4, 3, 3])})
    sd = G.load_state_dict(path, dtype=dtype)
    assert set(sd) == {"blk.weight", "blk.bias", "norm.scale", "conv.weight"}                      # prefix stripped, the un-prefixed tensor dropped
    want = (torch.from_numpy(d.copy()).to(dtype)[:, None] * torch.from_numpy(q.copy())).reshape(48, 64)      # the reference's d * q in the target dtype
    assert sd["blk.weight"].dtype == dtype and torch.equal(sd["blk.weight"], want)
    assert float((sd["blk.weight"].float() - torch.from_numpy(w)).abs().max()) <= 0.05 * 4 / 127 * 1.01 + 2e-3      # and it is the weight, to Q8_0 accuracy
    assert torch.equal(sd["blk.bias"], torch.from_numpy(bias).to(dtype))
    assert sd["conv.weight"].shape == (8, 4, 3, 3) and torch.equal(sd["conv.weight"], torch.from_numpy(conv).to(dtype))
    assert torch.equal(sd["norm.scale"], torch.from_numpy(norm).to(dtype))


def test_container_edge_cases(ldx, tmp_path):
    G = ldx.gguf_loader
    p = str(tmp_path / "a.gguf")
    _write(p, [("w", (2, 32), G.GGML_F16, np.arange(64, dtype=np.float16).tobytes())], {"general.architecture": (8, "llama")})
    with pytest.raises(ValueError, match="architecture"):                 # the reference refuses anything but flux / sd1 / sdxl / t5 (:618-621)
        G.load_state_dict(p)
    _write(p, [("w", (2, 32), G.GGML_F16, np.arange(64, dtype=np.float16).tobytes())], {}, version=2, align=32)
    sd = G.load_state_dict(p, dtype=None)                                 # no prefix anywhere: keys kept; dtype None: storage dtype kept
    assert list(sd) == ["w"] and sd["w"].dtype == torch.float16 and sd["w"].shape == (2, 32) and float(sd["w"][1, 31]) == 63.0
    _write(p, [("w", (2, 48), G.GGML_Q8_0, b"\0" * 102)], {})
    with pytest.raises(ValueError, match="multiple of 32"):
        G.load_state_dict(p)
    with open(p, "wb") as f:
        f.write(b"NOPE" + b"\0" * 64)
    with pytest.raises(ValueError, match="not a GGUF"):
        G.load_state_dict(p)
    _write(p, [("w", (4,), 12, b"\0" * 16)], {})                          # a K-quant: the reference has no dequantiser for it either
    with pytest.raises(NotImplementedError):
        G.load_state_dict(p)
