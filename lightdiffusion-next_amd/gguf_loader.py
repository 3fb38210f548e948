"""GGUF checkpoints (F32 / F16 / BF16 / Q8_0) -> a plain 16-bit state dict for the engines.

The reference ships Flux as `flux1-dev-Q8_0.gguf`: `Quantize/Quantizer.py:581-665` (`gguf_sd_loader`) reads it with the `gguf` package, strips the
`model.diffusion_model.` prefix, restores the torch shapes (GGUF stores dimensions innermost first; a `comfy.gguf.orig_shape.<name>` int32 array overrides),
and `GGMLOps.Linear` de-quantises every Q8_0 weight per forward (`:94-112`: blocks of 34 bytes = one fp16 scale d + 32 int8 values q, weight = d * q
computed in the target dtype).  ldx has no per-forward de-quantisation — weights live on the device in 16 bit (or are re-quantised once to MX fp8) — so the
file is de-quantised ONCE here, with the same arithmetic, into the state dict `FluxEngine` / `UNetEngine` / `T5Engine` take.

PARITY UNPINNED: the `gguf` package is not installed in the build image, so the reference's loader cannot be run against this one; the container layout
below is the published GGUF v2 / v3 format, the block arithmetic restates Quantizer.py:94-112, the key / shape / architecture handling restates :581-665.
Pure host code (numpy + torch on the CPU); nothing here runs in the hot path.
"""
from __future__ import annotations

import struct
from typing import Dict, Optional, Tuple

import numpy as np
import torch

GGUF_MAGIC = 0x46554747            # b"GGUF" little endian
GGML_F32, GGML_F16, GGML_Q8_0, GGML_BF16 = 0, 1, 8, 30
Q8_0_BLOCK, Q8_0_BYTES = 32, 34
_SCALARS = {0: "<B", 1: "<b", 2: "<H", 3: "<h", 4: "<I", 5: "<i", 6: "<f", 7: "<?", 10: "<Q", 11: "<q", 12: "<d"}
_ARCHS = {"flux", "sd1", "sdxl", "t5", "t5encoder"}       # Quantizer.py:618-621


class _Reader:
    def __init__(self, buf: memoryview):
        self.b, self.o = buf, 0

    def take(self, fmt: str):
        v = struct.unpack_from(fmt, self.b, self.o)[0]
        self.o += struct.calcsize(fmt)
        return v

    def string(self) -> str:
        n = self.take("<Q")
        s = bytes(self.b[self.o:self.o + n]).decode("utf-8")
        self.o += n
        return s

    def value(self, t: int):
        if t in _SCALARS:
            return self.take(_SCALARS[t])
        if t == 8:
            return self.string()
        if t == 9:
            et, n = self.take("<I"), self.take("<Q")
            return [self.value(et) for _ in range(n)]
        raise ValueError(f"GGUF: unknown metadata value type {t}")


def dequantize_q8_0(raw: torch.Tensor, shape, dtype: torch.dtype = torch.float16) -> torch.Tensor:
    """Quantizer.py:94-112 on a flat uint8 tensor of 34-byte blocks: d = fp16 scale -> dtype, q = int8, weight = d * q in `dtype`."""
    blocks = raw.reshape(-1, Q8_0_BYTES)
    d = blocks[:, :2].contiguous().view(torch.float16).to(dtype)
    q = blocks[:, 2:].contiguous().view(torch.int8)
    return (d * q).reshape(tuple(shape))


def read_gguf(path: str) -> Tuple[Dict[str, object], Dict[str, Tuple[tuple, int, np.ndarray]]]:
    """(metadata, tensors) of a GGUF v2 / v3 file; tensors[name] = (GGUF dims innermost first, ggml type, raw uint8 view of the mapped file)."""
    mm = np.memmap(path, dtype=np.uint8, mode="r")
    r = _Reader(memoryview(mm))
    if r.take("<I") != GGUF_MAGIC:
        raise ValueError(f"{path}: not a GGUF file")
    ver = r.take("<I")
    if ver not in (2, 3):
        raise ValueError(f"{path}: GGUF version {ver} (2 and 3 are supported)")
    n_t, n_kv = r.take("<Q"), r.take("<Q")
    meta: Dict[str, object] = {}
    for _ in range(n_kv):
        k = r.string()
        meta[k] = r.value(r.take("<I"))
    infos = []
    for _ in range(n_t):
        name = r.string()
        nd = r.take("<I")
        dims = tuple(r.take("<Q") for _ in range(nd))
        infos.append((name, dims, r.take("<I"), r.take("<Q")))
    align = int(meta.get("general.alignment", 32))
    base = (r.o + align - 1) // align * align
    tensors = {}
    for name, dims, typ, off in infos:
        n = int(np.prod(dims)) if dims else 1
        if typ == GGML_F32:
            nb = 4 * n
        elif typ in (GGML_F16, GGML_BF16):
            nb = 2 * n
        elif typ == GGML_Q8_0:
            if not dims or dims[0] % Q8_0_BLOCK:
                raise ValueError(f"{path}: {name}: Q8_0 needs an innermost dimension that is a multiple of 32, got {dims}")
            nb = n // Q8_0_BLOCK * Q8_0_BYTES
        else:
            raise NotImplementedError(f"{path}: {name}: ggml type {typ} (F32, F16, BF16 and Q8_0 are what the reference's dequantize_functions cover)")
        if base + off + nb > mm.size:
            raise ValueError(f"{path}: {name}: data runs past the end of the file")
        tensors[name] = (dims, typ, mm[base + off:base + off + nb])
    return meta, tensors


def load_state_dict(path: str, dtype: torch.dtype = torch.float16, handle_prefix: Optional[str] = "model.diffusion_model.") -> Dict[str, torch.Tensor]:
    """gguf_sd_loader + dequantize_tensor in one step: {key: dense tensor in `dtype`} (F32 tensors — biases, norms — stay fp32 as in the reference's
    TORCH_COMPATIBLE path only if dtype is None; with a dtype everything is cast).  Keys lose `handle_prefix` if any tensor carries it (the others are
    dropped, :593-605); the architecture string is checked against the reference's list (:607-621)."""
    meta, tensors = read_gguf(path)
    arch = meta.get("general.architecture")
    if arch is not None:
        if not isinstance(arch, str):
            raise TypeError(f"Bad type for GGUF general.architecture key: expected string, got {type(arch).__name__}")
        if arch not in _ARCHS:
            raise ValueError(f"Unexpected architecture type in GGUF file, expected one of flux, sd1, sdxl, t5encoder but got {arch!r}")
    has_prefix = handle_prefix is not None and any(n.startswith(handle_prefix) for n in tensors)
    out: Dict[str, torch.Tensor] = {}
    for name, (dims, typ, raw) in tensors.items():
        key = name
        if has_prefix:
            if not name.startswith(handle_prefix):
                continue
            key = name[len(handle_prefix):]
        orig = meta.get(f"comfy.gguf.orig_shape.{name}")
        shape = tuple(int(v) for v in orig) if orig is not None else tuple(int(v) for v in reversed(dims))
        t = torch.from_numpy(np.array(raw, copy=True))
        if typ == GGML_F32:
            x = t.view(torch.float32).reshape(shape)
        elif typ == GGML_F16:
            x = t.view(torch.float16).reshape(shape)
        elif typ == GGML_BF16:
            x = t.view(torch.bfloat16).reshape(shape)
        else:
            x = dequantize_q8_0(t, shape, dtype if dtype is not None else torch.float16)
        out[key] = x.to(dtype) if dtype is not None else x
    return out
