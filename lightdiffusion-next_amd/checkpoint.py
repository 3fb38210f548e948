"""Checkpoint ingestion for the engines: SD1.5 checkpoint splitting and UNet layout sniffing, and the LoRA
merge  W += strength * alpha / rank * (up @ down)  that the reference applies before the weights reach the accelerator hook.

All of this is load-time host work (the result is handed to ldx_load_tensor / UNetEngine); nothing here runs per step.
Reference: src/FileManaging/Loader.py:11-111 (CheckpointLoaderSimple), src/Model/LoRas.py:15-155 (key maps, load_lora),
src/Model/ModelPatcher.py:621-650 (calculate_weight), src/NeuralNetwork/unet.py:12-185 (diffusers <-> ldm key map).
"""
import os
import re
from typing import Dict, Optional, Sequence, Tuple, Union

import torch

from .weights import UNetConfig

UNET_PREFIX = "model.diffusion_model."
VAE_PREFIX = "first_stage_model."
CLIP_PREFIX = "cond_stage_model.transformer."


def split_sd15_checkpoint(sd: Dict[str, torch.Tensor]):
    """A full SD1.5 state dict (e.g. safetensors.torch.load_file) -> (unet, vae, clip) state dicts with the prefixes the
    reference strips (Loader.py:41-68): model.diffusion_model.*, first_stage_model.*, cond_stage_model.transformer.*
    (the CLIP engine strips the remaining "text_model." itself)."""
    unet = {k[len(UNET_PREFIX):]: v for k, v in sd.items() if k.startswith(UNET_PREFIX)}
    vae = {k[len(VAE_PREFIX):]: v for k, v in sd.items() if k.startswith(VAE_PREFIX)}
    clip = {k[len(CLIP_PREFIX):]: v for k, v in sd.items() if k.startswith(CLIP_PREFIX)}
    return unet, vae, clip


def detect_unet_config(unet_sd: Dict[str, torch.Tensor], num_heads: int = 8) -> UNetConfig:
    """Recover the UNetModel1 layout from tensor names and shapes (the reference's detect_unet_config, unet.py:773-1080, for
    the SD1.x family): model_channels from the first conv, channel_mult from the widths of the residual blocks, number of
    residual blocks and transformer depths by counting keys.  The head count is not recorded in the weights (8 for SD1.x)."""
    mc = unet_sd["input_blocks.0.0.weight"].shape[0]
    in_ch = unet_sd["input_blocks.0.0.weight"].shape[1]
    out_ch = unet_sd["out.2.weight"].shape[0]
    n_in = 1 + max(int(k.split(".")[1]) for k in unet_sd if k.startswith("input_blocks."))
    mult, nres, tdepth = [], [], []
    cur_res, ib = 0, 1
    while ib < n_in:
        if f"input_blocks.{ib}.0.op.weight" in unet_sd:          # Downsample closes a level
            nres.append(cur_res)
            cur_res = 0
        else:
            width = unet_sd[f"input_blocks.{ib}.0.out_layers.3.weight"].shape[0]
            if cur_res == 0:
                mult.append(width // mc)
            cur_res += 1
            depth = 0
            while f"input_blocks.{ib}.1.transformer_blocks.{depth}.norm1.weight" in unet_sd:
                depth += 1
            tdepth.append(depth)
        ib += 1
    nres.append(cur_res)
    n_out = 1 + max(int(k.split(".")[1]) for k in unet_sd if k.startswith("output_blocks."))
    tdepth_out = []
    for ob in range(n_out):
        depth = 0
        while f"output_blocks.{ob}.1.transformer_blocks.{depth}.norm1.weight" in unet_sd:
            depth += 1
        tdepth_out.append(depth)
    mid = 0
    while f"middle_block.1.transformer_blocks.{mid}.norm1.weight" in unet_sd:
        mid += 1
    ctx = next(v.shape[1] for k, v in unet_sd.items() if k.endswith("attn2.to_k.weight"))
    return UNetConfig(in_channels=in_ch, out_channels=out_ch, model_channels=mc, channel_mult=tuple(mult), num_res_blocks=tuple(nres),
                      transformer_depth=tuple(tdepth),
                      transformer_depth_output=tuple(reversed(tdepth_out)),      # consumed with .pop() from the end (unet.py:561)
                      transformer_depth_middle=mid,
                      num_heads=num_heads, context_dim=ctx)


# ------------------------------------------------------------------------------------------------------
_RES = {"in_layers.2": "conv1", "emb_layers.1": "time_emb_proj", "out_layers.3": "conv2", "skip_connection": "conv_shortcut",
        "in_layers.0": "norm1", "out_layers.0": "norm2"}
_ATT = ("proj_in", "proj_out", "norm")
_TB = ("norm1", "norm2", "norm3", "attn1.to_q", "attn1.to_k", "attn1.to_v", "attn1.to_out.0", "attn2.to_q", "attn2.to_k", "attn2.to_v",
       "attn2.to_out.0", "ff.net.0.proj", "ff.net.2")
_BASIC = {"conv_in": "input_blocks.0.0", "conv_norm_out": "out.0", "conv_out": "out.2", "time_embedding.linear_1": "time_embed.0",
          "time_embedding.linear_2": "time_embed.2"}


def unet_to_diffusers(cfg: UNetConfig) -> Dict[str, str]:
    """diffusers module path -> ldm module path (no .weight/.bias suffix), the table unet.unet_to_diffusers builds
    (unet.py:85-185) for the blocks an SD1.x UNet has."""
    m = {}
    td, tdo = list(cfg.transformer_depth), list(cfg.transformer_depth_output)
    nb = len(cfg.channel_mult)
    for x in range(nb):
        n = 1 + (cfg.num_res_blocks[x] + 1) * x
        for i in range(cfg.num_res_blocks[x]):
            for a, b in _RES.items():
                m[f"down_blocks.{x}.resnets.{i}.{b}"] = f"input_blocks.{n}.0.{a}"
            depth = td.pop(0)
            if depth > 0:
                for a in _ATT:
                    m[f"down_blocks.{x}.attentions.{i}.{a}"] = f"input_blocks.{n}.1.{a}"
                for t in range(depth):
                    for a in _TB:
                        m[f"down_blocks.{x}.attentions.{i}.transformer_blocks.{t}.{a}"] = f"input_blocks.{n}.1.transformer_blocks.{t}.{a}"
            n += 1
        m[f"down_blocks.{x}.downsamplers.0.conv"] = f"input_blocks.{n}.0.op"
    for a in _ATT:
        m[f"mid_block.attentions.0.{a}"] = f"middle_block.1.{a}"
    for t in range(cfg.transformer_depth_middle):
        for a in _TB:
            m[f"mid_block.attentions.0.transformer_blocks.{t}.{a}"] = f"middle_block.1.transformer_blocks.{t}.{a}"
    for i, n in enumerate((0, 2)):
        for a, b in _RES.items():
            m[f"mid_block.resnets.{i}.{b}"] = f"middle_block.{n}.{a}"
    nres_rev = list(reversed(cfg.num_res_blocks))
    for x in range(nb):
        n = (nres_rev[x] + 1) * x
        length = nres_rev[x] + 1
        for i in range(length):
            c = 1
            for a, b in _RES.items():
                m[f"up_blocks.{x}.resnets.{i}.{b}"] = f"output_blocks.{n}.0.{a}"
            depth = tdo.pop()
            if depth > 0:
                c += 1
                for a in _ATT:
                    m[f"up_blocks.{x}.attentions.{i}.{a}"] = f"output_blocks.{n}.1.{a}"
                for t in range(depth):
                    for a in _TB:
                        m[f"up_blocks.{x}.attentions.{i}.transformer_blocks.{t}.{a}"] = f"output_blocks.{n}.1.transformer_blocks.{t}.{a}"
            if i == length - 1:
                m[f"up_blocks.{x}.upsamplers.0.conv"] = f"output_blocks.{n}.{c}.conv"
            n += 1
    m.update(_BASIC)
    return m


def lora_key_map_unet(cfg: UNetConfig, unet_keys) -> Dict[str, str]:
    """LoRA module name -> UNet weight key (engine naming, no diffusion_model. prefix): model_lora_keys_unet (LoRas.py:88-121) —
    the ldm-style 'lora_unet_<path with _>' names for every weight, plus the diffusers-style names kohya / diffusers LoRAs use."""
    keys = set(unet_keys)
    out = {}
    for k in keys:
        if k.endswith(".weight"):
            out["lora_unet_" + k[:-len(".weight")].replace(".", "_")] = k
    for dpath, lpath in unet_to_diffusers(cfg).items():
        wk = lpath + ".weight"
        if wk not in keys:
            continue
        out["lora_unet_" + dpath.replace(".", "_")] = wk
        for pre in ("", "unet."):
            dk = pre + dpath.replace(".to_", ".processor.to_")
            if dk.endswith(".to_out.0"):
                dk = dk[:-2]
            out[dk] = wk
    return out


_CLIP_MAP = {"mlp.fc1": "mlp_fc1", "mlp.fc2": "mlp_fc2", "self_attn.k_proj": "self_attn_k_proj", "self_attn.q_proj": "self_attn_q_proj",
             "self_attn.v_proj": "self_attn_v_proj", "self_attn.out_proj": "self_attn_out_proj"}


def lora_key_map_clip(clip_keys) -> Dict[str, str]:
    """model_lora_keys_clip (LoRas.py:58-85) for a CLIP-L state dict with 'text_model.' keys."""
    keys, out = set(clip_keys), {}
    for k in keys:
        mm = re.match(r"text_model\.encoder\.layers\.(\d+)\.(.+)\.weight$", k)
        if mm and mm.group(2) in _CLIP_MAP:
            b, c = mm.group(1), mm.group(2)
            out[f"lora_te_text_model_encoder_layers_{b}_{_CLIP_MAP[c]}"] = k
            out[f"lora_te1_text_model_encoder_layers_{b}_{_CLIP_MAP[c]}"] = k
            out[f"text_encoder.text_model.encoder.layers.{b}.{c}"] = k
    return out


def merge_lora(weights: Dict[str, torch.Tensor], lora: Dict[str, torch.Tensor], key_map: Dict[str, str], strength: float = 1.0) -> Tuple[Dict[str, torch.Tensor], int]:
    """load_lora (LoRas.py:15-55) + ModelPatcher.calculate_weight (ModelPatcher.py:621-650): for every LoRA module x in key_map
    with x.lora_up.weight / x.lora_down.weight:  W += (strength * alpha / rank) * (up.flatten(1) @ down.flatten(1)).reshape(W.shape),
    computed in fp32 and cast back to W's dtype; alpha defaults to rank (scale 1).  Returns (new dict, number of patched tensors)."""
    out = dict(weights)
    n = 0
    for x, target in key_map.items():
        up, down = lora.get(f"{x}.lora_up.weight"), lora.get(f"{x}.lora_down.weight")
        if up is None or down is None or target not in out:
            continue
        a = float(strength)
        alpha = lora.get(f"{x}.alpha")
        if alpha is not None:
            a *= float(alpha) / down.shape[0]
        w = out[target]
        delta = a * torch.mm(up.float().flatten(start_dim=1), down.float().flatten(start_dim=1))
        out[target] = (w.float() + delta.reshape(w.shape)).to(w.dtype) if w.dtype != torch.float32 else w + delta.reshape(w.shape)
        n += 1
    return out, n


# ---------------------------------------------------------------------------------------------------
# Textual-inversion embedding files (src/SD15/SDToken.py:108-206)
_EMBED_EXT = (".safetensors", ".pt", ".bin")


def _embedding_search_dirs(directories) -> list:
    """Every directory under the given ones, symlinks followed (expand_directory_list, SDToken.py:108-122)."""
    out = set()
    for d in directories:
        out.add(d)
        for root, _sub, _files in os.walk(d, followlinks=True):
            out.add(root)
    return list(out)


def load_embed(embedding_name: str, embedding_directory: Union[str, Sequence[str]], embedding_size: int,
               embed_key: Optional[str] = None) -> Optional[torch.Tensor]:
    """The tensor the reference's load_embed returns for "embedding:<name>" (SDToken.py:125-206), or None.

    File lookup: <dir>/<name> for every directory of the tree, else the same path with .safetensors / .pt / .bin appended;
    a name that escapes its directory ("../x") is refused.  Layouts, in the reference's order of precedence:
      {"string_to_param": {tok: T}}   -> first T (A1111 .pt)
      [ {k: T, ...}, ... ]            -> rows of every T whose last dim == embedding_size, concatenated
      {embed_key: T, ...}             -> T            (e.g. "clip_l" of an SDXL-style file)
      {k: T, ...}                     -> first T      (plain safetensors: "emb_params")
    A file that fails to load is skipped (None), as the reference does after logging."""
    dirs = [embedding_directory] if isinstance(embedding_directory, str) else list(embedding_directory)
    found = None
    for d in _embedding_search_dirs(dirs):
        base, path = os.path.abspath(d), os.path.abspath(os.path.join(d, embedding_name))
        try:
            if os.path.commonpath((base, path)) != base:
                continue
        except ValueError:
            continue
        if os.path.isfile(path):
            found = path
        else:
            found = next((path + e for e in _EMBED_EXT if os.path.isfile(path + e)), None)
        if found is not None:
            break
    if found is None:
        return None
    try:
        if found.lower().endswith(".safetensors"):
            import safetensors.torch
            data = safetensors.torch.load_file(found, device="cpu")
        else:
            data = torch.load(found, weights_only=True, map_location="cpu")
    except Exception:
        return None
    if isinstance(data, dict) and "string_to_param" in data:
        return next(iter(data["string_to_param"].values()))
    if isinstance(data, list):
        rows = [t.reshape(-1, t.shape[-1]) for entry in data for t in entry.values() if t.shape[-1] == embedding_size]
        return torch.cat(rows, dim=0)
    if embed_key is not None and embed_key in data:
        return data[embed_key]
    return next(iter(data.values()))


class EmbeddingDirectory:
    """`embeddings=` argument of prompt.tokenize_with_weights backed by files: .get(name) = load_embed(name, ...), like
    SDTokenizer._try_get_embedding (SDToken.py:264-290; the retry without trailing commas is done by the caller)."""

    def __init__(self, directories, embedding_size: int = 768, embed_key: str = "clip_l"):
        self.directories, self.embedding_size, self.embed_key = directories, embedding_size, embed_key

    def get(self, name: str):
        return load_embed(name, self.directories, self.embedding_size, self.embed_key)
