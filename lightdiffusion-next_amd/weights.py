"""SD1.5-layout UNet state dict: key/shape inventory and seeded synthetic weights.

No checkpoint exists offline, so parity fixtures and benchmarks use synthetic weights
written into the *exact* key layout the reference builds in UNetModel1.__init__
(src/NeuralNetwork/unet.py:333-677; canonical SD1.5 config = SURVEY.md Appendix B: 686 keys,
859 520 964 parameters).  The inventory below is derived from the config alone, so it can be checked
against the reference module's own state_dict (oracle/ref_capture.py does, strictly).
"""
import math
import zlib
from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import torch


@dataclass
class UNetConfig:
    """Keyword arguments of UNetModel1 that the SD1.5 family uses (unet.py:208-252, SD15.py:17-28)."""

    in_channels: int = 4
    out_channels: int = 4
    model_channels: int = 320
    channel_mult: Tuple[int, ...] = (1, 2, 4, 4)
    num_res_blocks: Tuple[int, ...] = (2, 2, 2, 2)
    transformer_depth: Tuple[int, ...] = (1, 1, 1, 1, 1, 1, 0, 0)
    transformer_depth_output: Tuple[int, ...] = (1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0)
    transformer_depth_middle: int = 1
    num_heads: int = 8
    context_dim: int = 768

    @staticmethod
    def sd15() -> "UNetConfig":
        return UNetConfig()

    @staticmethod
    def tiny(model_channels: int = 64, context_dim: int = 128) -> "UNetConfig":
        """Same topology as SD1.5, narrower: head dims 8/16/32 at model_channels=64."""
        return UNetConfig(model_channels=model_channels, context_dim=context_dim)

    def reference_kwargs(self) -> dict:
        """The dict SURVEY.md Appendix B feeds to SD15.sm_SD15 / UNetModel1."""
        return dict(
            use_checkpoint=False, image_size=32, use_spatial_transformer=True, legacy=False,
            adm_in_channels=None, in_channels=self.in_channels, out_channels=self.out_channels,
            model_channels=self.model_channels, num_res_blocks=list(self.num_res_blocks),
            transformer_depth=list(self.transformer_depth),
            transformer_depth_output=list(self.transformer_depth_output),
            channel_mult=list(self.channel_mult), transformer_depth_middle=self.transformer_depth_middle,
            use_linear_in_transformer=False, context_dim=self.context_dim,
            use_temporal_resblock=False, use_temporal_attention=False,
        )


def _res(spec, pre, cin, cout, ted):
    spec += [(f"{pre}.in_layers.0.weight", (cin,)), (f"{pre}.in_layers.0.bias", (cin,)),
             (f"{pre}.in_layers.2.weight", (cout, cin, 3, 3)), (f"{pre}.in_layers.2.bias", (cout,)),
             (f"{pre}.emb_layers.1.weight", (cout, ted)), (f"{pre}.emb_layers.1.bias", (cout,)),
             (f"{pre}.out_layers.0.weight", (cout,)), (f"{pre}.out_layers.0.bias", (cout,)),
             (f"{pre}.out_layers.3.weight", (cout, cout, 3, 3)), (f"{pre}.out_layers.3.bias", (cout,))]
    if cin != cout:
        spec += [(f"{pre}.skip_connection.weight", (cout, cin, 1, 1)), (f"{pre}.skip_connection.bias", (cout,))]


def _xf(spec, pre, c, depth, ctx):
    spec += [(f"{pre}.norm.weight", (c,)), (f"{pre}.norm.bias", (c,)),
             (f"{pre}.proj_in.weight", (c, c, 1, 1)), (f"{pre}.proj_in.bias", (c,))]
    for d in range(depth):
        b = f"{pre}.transformer_blocks.{d}"
        spec += [(f"{b}.attn1.to_q.weight", (c, c)), (f"{b}.attn1.to_k.weight", (c, c)),
                 (f"{b}.attn1.to_v.weight", (c, c)),
                 (f"{b}.attn1.to_out.0.weight", (c, c)), (f"{b}.attn1.to_out.0.bias", (c,)),
                 (f"{b}.ff.net.0.proj.weight", (8 * c, c)), (f"{b}.ff.net.0.proj.bias", (8 * c,)),
                 (f"{b}.ff.net.2.weight", (c, 4 * c)), (f"{b}.ff.net.2.bias", (c,)),
                 (f"{b}.attn2.to_q.weight", (c, c)), (f"{b}.attn2.to_k.weight", (c, ctx)),
                 (f"{b}.attn2.to_v.weight", (c, ctx)),
                 (f"{b}.attn2.to_out.0.weight", (c, c)), (f"{b}.attn2.to_out.0.bias", (c,)),
                 (f"{b}.norm1.weight", (c,)), (f"{b}.norm1.bias", (c,)),
                 (f"{b}.norm2.weight", (c,)), (f"{b}.norm2.bias", (c,)),
                 (f"{b}.norm3.weight", (c,)), (f"{b}.norm3.bias", (c,))]
    spec += [(f"{pre}.proj_out.weight", (c, c, 1, 1)), (f"{pre}.proj_out.bias", (c,))]


def unet_state_dict_spec(cfg: UNetConfig) -> List[Tuple[str, Tuple[int, ...]]]:
    """(key, shape) for every tensor, walking the structure like UNetModel1.__init__ (unet.py:333-677)."""
    mc, ted, ctx = cfg.model_channels, 4 * cfg.model_channels, cfg.context_dim
    spec: List[Tuple[str, Tuple[int, ...]]] = [
        ("time_embed.0.weight", (ted, mc)), ("time_embed.0.bias", (ted,)),
        ("time_embed.2.weight", (ted, ted)), ("time_embed.2.bias", (ted,)),
        ("input_blocks.0.0.weight", (mc, cfg.in_channels, 3, 3)), ("input_blocks.0.0.bias", (mc,)),
    ]
    td = list(cfg.transformer_depth)
    tdo = list(cfg.transformer_depth_output)
    ch, ib, chans = mc, 1, [mc]
    nl = len(cfg.channel_mult)
    for level, mult in enumerate(cfg.channel_mult):
        for _ in range(cfg.num_res_blocks[level]):
            _res(spec, f"input_blocks.{ib}.0", ch, mult * mc, ted)
            ch = mult * mc
            depth = td.pop(0)
            if depth > 0:
                _xf(spec, f"input_blocks.{ib}.1", ch, depth, ctx)
            chans.append(ch)
            ib += 1
        if level != nl - 1:
            spec += [(f"input_blocks.{ib}.0.op.weight", (ch, ch, 3, 3)), (f"input_blocks.{ib}.0.op.bias", (ch,))]
            chans.append(ch)
            ib += 1
    _res(spec, "middle_block.0", ch, ch, ted)
    if cfg.transformer_depth_middle >= 0:
        _xf(spec, "middle_block.1", ch, cfg.transformer_depth_middle, ctx)
        _res(spec, "middle_block.2", ch, ch, ted)
    ob = 0
    for level in reversed(range(nl)):
        mult = cfg.channel_mult[level]
        for i in range(cfg.num_res_blocks[level] + 1):
            ich = chans.pop()
            _res(spec, f"output_blocks.{ob}.0", ch + ich, mc * mult, ted)
            ch = mc * mult
            sub = 1
            depth = tdo.pop()
            if depth > 0:
                _xf(spec, f"output_blocks.{ob}.1", ch, depth, ctx)
                sub += 1
            if level and i == cfg.num_res_blocks[level]:
                spec += [(f"output_blocks.{ob}.{sub}.conv.weight", (ch, ch, 3, 3)),
                         (f"output_blocks.{ob}.{sub}.conv.bias", (ch,))]
            ob += 1
    spec += [("out.0.weight", (ch,)), ("out.0.bias", (ch,)),
             ("out.2.weight", (cfg.out_channels, mc, 3, 3)), ("out.2.bias", (cfg.out_channels,))]
    return spec


def _is_norm_weight(key: str) -> bool:
    parts = key.split(".")
    if parts[-1] != "weight":
        return False
    name = ".".join(parts[:-1])
    return (name.endswith("in_layers.0") or name.endswith("out_layers.0") or name.endswith(".norm")
            or name.endswith("norm1") or name.endswith("norm2") or name.endswith("norm3")
            or name == "out.0" or name.endswith("layer_norm1") or name.endswith("layer_norm2")
            or name.endswith("final_layer_norm") or ("norm" in parts[-2]))


_is_norm_weight_base = _is_norm_weight


def synth_tensor(key: str, shape, seed: int, dtype=torch.float16) -> torch.Tensor:
    """Deterministic per-key fill (independent of iteration order): biases ~N(0,.02), norm scales
    1+N(0,.02), everything else N(0, 1/fan_in) — variance-preserving so activations stay O(1)."""
    g = torch.Generator(device="cpu")
    g.manual_seed((seed * 1000003 + zlib.crc32(key.encode())) & 0x7FFFFFFF)
    n = torch.randn(tuple(shape), generator=g, dtype=torch.float32)
    if key.endswith(".bias"):
        t = 0.02 * n
    elif _is_norm_weight(key) or key.endswith("_norm.scale"):
        t = 1.0 + 0.02 * n
    else:
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        t = n / math.sqrt(max(fan_in, 1))
        # T5 attention is unscaled (no 1/sqrt(d)): trained q/k projections are small; 0.4 ~ d^(-1/4) keeps the logits O(1)
        if key.endswith("SelfAttention.q.weight") or key.endswith("SelfAttention.k.weight"):
            t = 0.4 * t
    return t.to(dtype)


def synth_state_dict(spec, seed: int = 1234, dtype=torch.float16) -> Dict[str, torch.Tensor]:
    return {k: synth_tensor(k, s, seed, dtype) for k, s in spec}


# ---- one synthetic state dict per NODE (multi-GPU start-up): rank 0 builds it once into a file under /dev/shm, the other ranks map that file.
# Eight ranks x 859.5 M parameters synthesised in parallel cost 30-40 s of host work each and fight for one socket's memory bandwidth; the mapped
# copy costs the page-cache read only.  Layout: tensors in spec order at 64-byte aligned offsets, raw little-endian elements, no header — both
# sides derive every offset from (spec, dtype), so the file holds data only.
def state_dict_layout(spec, dtype=torch.float16):
    """[(key, shape, byte offset, byte length)], total bytes."""
    esz = torch.empty((), dtype=dtype).element_size()
    out, off = [], 0
    for k, shp in spec:
        n = 1
        for d in shp:
            n *= d
        out.append((k, tuple(shp), off, n * esz))
        off = (off + n * esz + 63) // 64 * 64
    return out, max(off, 64)


def _map_state_dict(path, spec, dtype, create):
    layout, total = state_dict_layout(spec, dtype)
    buf = torch.from_file(path, shared=True, size=total, dtype=torch.uint8) if create else torch.from_file(path, shared=False, size=total, dtype=torch.uint8)
    return {k: buf[off:off + nb].view(dtype).reshape(shp) for k, shp, off, nb in layout}, buf


def publish_state_dict(spec, path: str, seed: int = 1234, dtype=torch.float16) -> Dict[str, torch.Tensor]:
    """Build synth_state_dict(spec, seed, dtype) INTO a file at `path` (written under a temporary name, then renamed: a reader never maps a partial
    file) and return it as views of the mapping.  Bit-identical to synth_state_dict."""
    import os
    tmp = f"{path}.tmp.{os.getpid()}"
    try:
        sd, buf = _map_state_dict(tmp, spec, dtype, create=True)
        for k, shp in spec:
            sd[k].copy_(synth_tensor(k, shp, seed, dtype))
        del sd, buf                                 # unmap: the shared mapping's pages are the file's pages
        os.replace(tmp, path)
    finally:
        # a failed build must not leave gigabytes behind in tmpfs (the caller falls back to private synthesis and carries on)
        try:
            os.unlink(tmp)
        except FileNotFoundError:
            pass
    return attach_state_dict(spec, path, dtype)


def attach_state_dict(spec, path: str, dtype=torch.float16) -> Dict[str, torch.Tensor]:
    """Map a file written by publish_state_dict (private copy-on-write mapping: nothing a rank does to its tensors reaches the others)."""
    import os
    _, total = state_dict_layout(spec, dtype)
    have = os.path.getsize(path)
    if have != total:
        raise RuntimeError(f"shared state dict {path}: {have} bytes, expected {total} (another model / dtype, or a stale file)")
    sd, _ = _map_state_dict(path, spec, dtype, create=False)
    return sd


def param_count(spec) -> int:
    n = 0
    for _, s in spec:
        m = 1
        for d in s:
            m *= d
        n += m
    return n


# ------------------------------------------------------------------------------------------------------
@dataclass
class VAEConfig:
    """ddconfig of the SD1.5 AutoencoderKL (src/AutoEncoders/VariationalAE.py:612-640)."""

    z_channels: int = 4
    ch: int = 128
    ch_mult: Tuple[int, ...] = (1, 2, 4, 4)
    num_res_blocks: int = 2
    out_ch: int = 3
    use_post_quant: bool = True

    @staticmethod
    def tiny(ch: int = 64) -> "VAEConfig":
        return VAEConfig(ch=ch)


def vae_decoder_state_dict_spec(cfg: VAEConfig):
    """Decoder + post_quant_conv keys in the order Decoder.__init__ creates them (VariationalAE.py:416-530)."""
    spec = []

    def res(pre, cin, cout):
        nonlocal spec
        spec += [(f"{pre}.norm1.weight", (cin,)), (f"{pre}.norm1.bias", (cin,)),
                 (f"{pre}.conv1.weight", (cout, cin, 3, 3)), (f"{pre}.conv1.bias", (cout,)),
                 (f"{pre}.norm2.weight", (cout,)), (f"{pre}.norm2.bias", (cout,)),
                 (f"{pre}.conv2.weight", (cout, cout, 3, 3)), (f"{pre}.conv2.bias", (cout,))]
        if cin != cout:
            spec += [(f"{pre}.nin_shortcut.weight", (cout, cin, 1, 1)), (f"{pre}.nin_shortcut.bias", (cout,))]

    nl = len(cfg.ch_mult)
    block_in = cfg.ch * cfg.ch_mult[-1]
    spec += [("decoder.conv_in.weight", (block_in, cfg.z_channels, 3, 3)), ("decoder.conv_in.bias", (block_in,))]
    res("decoder.mid.block_1", block_in, block_in)
    spec += [("decoder.mid.attn_1.norm.weight", (block_in,)), ("decoder.mid.attn_1.norm.bias", (block_in,))]
    for n in ("q", "k", "v", "proj_out"):
        spec += [(f"decoder.mid.attn_1.{n}.weight", (block_in, block_in, 1, 1)), (f"decoder.mid.attn_1.{n}.bias", (block_in,))]
    res("decoder.mid.block_2", block_in, block_in)
    for lv in reversed(range(nl)):
        block_out = cfg.ch * cfg.ch_mult[lv]
        for i in range(cfg.num_res_blocks + 1):
            res(f"decoder.up.{lv}.block.{i}", block_in, block_out)
            block_in = block_out
        if lv != 0:
            spec += [(f"decoder.up.{lv}.upsample.conv.weight", (block_in, block_in, 3, 3)),
                     (f"decoder.up.{lv}.upsample.conv.bias", (block_in,))]
    spec += [("decoder.norm_out.weight", (block_in,)), ("decoder.norm_out.bias", (block_in,)),
             ("decoder.conv_out.weight", (cfg.out_ch, block_in, 3, 3)), ("decoder.conv_out.bias", (cfg.out_ch,))]
    if cfg.use_post_quant:
        spec += [("post_quant_conv.weight", (cfg.z_channels, cfg.z_channels, 1, 1)), ("post_quant_conv.bias", (cfg.z_channels,))]
    return spec


def vae_encoder_state_dict_spec(cfg: VAEConfig, in_channels: int = 3):
    """Encoder + quant_conv keys in the order Encoder.__init__ creates them (VariationalAE.py:257-377)."""
    spec = [("encoder.conv_in.weight", (cfg.ch, in_channels, 3, 3)), ("encoder.conv_in.bias", (cfg.ch,))]

    def res(pre, cin, cout):
        nonlocal spec
        spec += [(f"{pre}.norm1.weight", (cin,)), (f"{pre}.norm1.bias", (cin,)),
                 (f"{pre}.conv1.weight", (cout, cin, 3, 3)), (f"{pre}.conv1.bias", (cout,)),
                 (f"{pre}.norm2.weight", (cout,)), (f"{pre}.norm2.bias", (cout,)),
                 (f"{pre}.conv2.weight", (cout, cout, 3, 3)), (f"{pre}.conv2.bias", (cout,))]
        if cin != cout:
            spec += [(f"{pre}.nin_shortcut.weight", (cout, cin, 1, 1)), (f"{pre}.nin_shortcut.bias", (cout,))]

    nl = len(cfg.ch_mult)
    block_in = cfg.ch
    for lv in range(nl):
        block_out = cfg.ch * cfg.ch_mult[lv]
        for i in range(cfg.num_res_blocks):
            res(f"encoder.down.{lv}.block.{i}", block_in, block_out)
            block_in = block_out
        if lv != nl - 1:
            spec += [(f"encoder.down.{lv}.downsample.conv.weight", (block_in, block_in, 3, 3)),
                     (f"encoder.down.{lv}.downsample.conv.bias", (block_in,))]
    res("encoder.mid.block_1", block_in, block_in)
    spec += [("encoder.mid.attn_1.norm.weight", (block_in,)), ("encoder.mid.attn_1.norm.bias", (block_in,))]
    for n in ("q", "k", "v", "proj_out"):
        spec += [(f"encoder.mid.attn_1.{n}.weight", (block_in, block_in, 1, 1)), (f"encoder.mid.attn_1.{n}.bias", (block_in,))]
    res("encoder.mid.block_2", block_in, block_in)
    spec += [("encoder.norm_out.weight", (block_in,)), ("encoder.norm_out.bias", (block_in,)),
             ("encoder.conv_out.weight", (2 * cfg.z_channels, block_in, 3, 3)), ("encoder.conv_out.bias", (2 * cfg.z_channels,))]
    if cfg.use_post_quant:
        spec += [("quant_conv.weight", (2 * cfg.z_channels, 2 * cfg.z_channels, 1, 1)), ("quant_conv.bias", (2 * cfg.z_channels,))]
    return spec


def vae_state_dict_spec(cfg: VAEConfig):
    """Full AutoencodingEngine state dict: encoder, decoder, quant / post_quant convs."""
    return vae_encoder_state_dict_spec(cfg) + vae_decoder_state_dict_spec(cfg)


@dataclass
class CLIPConfig:
    """include/clip/sd1_clip_config.json as read by CLIPTextModel_ (src/clip/CLIPTextModel.py:3-50)."""

    hidden_size: int = 768
    num_layers: int = 12
    num_heads: int = 12
    intermediate_size: int = 3072
    max_positions: int = 77
    vocab_size: int = 49408
    eos_token_id: int = 2          # the include/ json says 2 (not 49407): pooled output quirk, SURVEY A-9

    @staticmethod
    def tiny() -> "CLIPConfig":
        return CLIPConfig(hidden_size=128, num_layers=3, num_heads=2, intermediate_size=256, vocab_size=49408)


def clip_state_dict_spec(cfg: CLIPConfig):
    """text_model.* keys (prefix stripped) of CLIPTextModel_ (CLIPTextModel.py:25-50, Clip.py:14-294)."""
    e, f = cfg.hidden_size, cfg.intermediate_size
    spec = [("embeddings.token_embedding.weight", (cfg.vocab_size, e)),
            ("embeddings.position_embedding.weight", (cfg.max_positions, e))]
    for l in range(cfg.num_layers):
        p = f"encoder.layers.{l}"
        spec += [(f"{p}.layer_norm1.weight", (e,)), (f"{p}.layer_norm1.bias", (e,))]
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            spec += [(f"{p}.self_attn.{n}.weight", (e, e)), (f"{p}.self_attn.{n}.bias", (e,))]
        spec += [(f"{p}.layer_norm2.weight", (e,)), (f"{p}.layer_norm2.bias", (e,)),
                 (f"{p}.mlp.fc1.weight", (f, e)), (f"{p}.mlp.fc1.bias", (f,)),
                 (f"{p}.mlp.fc2.weight", (e, f)), (f"{p}.mlp.fc2.bias", (e,))]
    spec += [("final_layer_norm.weight", (e,)), ("final_layer_norm.bias", (e,))]
    spec += [("text_projection.weight", (e, e))]       # CLIPTextModel.text_projection (CLIPTextModel.py:126-128), no prefix
    return spec


# ------------------------------------------------------------------------------------------------------
@dataclass
class ESRGANConfig:
    """RRDBNet as the reference builds it (src/UltimateSDUpscale/RDRB.py:216-378): nf 64, gc 32, `num_blocks` RRDBs,
    log2(scale) upconv stages; defaults = RealESRGAN_x4plus / ESRGAN x4."""

    in_nc: int = 3
    out_nc: int = 3
    nf: int = 64
    gc: int = 32
    num_blocks: int = 23
    scale: int = 4

    @staticmethod
    def tiny() -> "ESRGANConfig":
        return ESRGANConfig(num_blocks=2)


def esrgan_state_dict_spec(cfg: ESRGANConfig):
    """RRDBNet's own ("old arch") module keys in creation order (RDRB.py:300-378)."""
    nf, gc, nu = cfg.nf, cfg.gc, int(math.log2(cfg.scale))
    spec = [("model.0.weight", (nf, cfg.in_nc, 3, 3)), ("model.0.bias", (nf,))]
    for i in range(cfg.num_blocks):
        for k in (1, 2, 3):
            for j in range(5):
                p = f"model.1.sub.{i}.RDB{k}.conv{j + 1}.0"
                co = gc if j < 4 else nf
                spec += [(p + ".weight", (co, nf + j * gc, 3, 3)), (p + ".bias", (co,))]
    spec += [(f"model.1.sub.{cfg.num_blocks}.weight", (nf, nf, 3, 3)), (f"model.1.sub.{cfg.num_blocks}.bias", (nf,))]
    for u in range(nu):
        spec += [(f"model.{3 * (u + 1)}.weight", (nf, nf, 3, 3)), (f"model.{3 * (u + 1)}.bias", (nf,))]
    spec += [(f"model.{3 * nu + 2}.weight", (nf, nf, 3, 3)), (f"model.{3 * nu + 2}.bias", (nf,)),
             (f"model.{3 * nu + 4}.weight", (cfg.out_nc, nf, 3, 3)), (f"model.{3 * nu + 4}.bias", (cfg.out_nc,))]
    return spec


def esrgan_new_to_old_arch(state):
    """RRDBNet.new_to_old_arch (RDRB.py:381-441): Real-ESRGAN / BSRGAN key names -> the module's own names.  A state dict
    already in the old layout is returned unchanged."""
    import re
    if any(k.startswith("model.") for k in state):
        return dict(state)
    out = {}
    nb = 1 + max(int(m.group(1)) for k in state for m in [re.match(r"(?:RRDB_trunk|body)\.(\d+)\.", k)] if m)
    ren = {"conv_first": "model.0", "trunk_conv": f"model.1.sub.{nb}", "conv_body": f"model.1.sub.{nb}"}
    max_up = 0
    for k, v in state.items():
        base, kind = k.rsplit(".", 1)
        m = re.match(r"(?:RRDB_trunk|body)\.(\d+)\.(?:RDB|rdb)(\d)\.conv(\d+)$", base)
        u = re.match(r"(?:upconv|conv_up)(\d)$", base)
        if base in ren:
            out[f"{ren[base]}.{kind}"] = v
        elif m:
            out[f"model.1.sub.{m.group(1)}.RDB{m.group(2)}.conv{m.group(3)}.0.{kind}"] = v
        elif u:
            out[f"model.{int(u.group(1)) * 3}.{kind}"] = v
            max_up = max(max_up, int(u.group(1)) * 3)
    for k, v in state.items():
        base, kind = k.rsplit(".", 1)
        if base in ("HRconv", "conv_hr"):
            out[f"model.{max_up + 2}.{kind}"] = v
        elif base == "conv_last":
            out[f"model.{max_up + 4}.{kind}"] = v
    return out


# ------------------------------------------------------------------------------------------------------
@dataclass
class T5Config:
    """src/clip/clip/t5_config_xxl.json as read by T5 (src/clip/FluxClip.py:476-519); defaults = T5-XXL encoder."""

    d_model: int = 4096
    d_ff: int = 10240
    num_layers: int = 24
    num_heads: int = 64
    vocab_size: int = 32128
    num_buckets: int = 32          # T5Attention.relative_attention_num_buckets (FluxClip.py:140)
    max_distance: int = 128        # .relative_attention_max_distance (:141)

    @staticmethod
    def tiny() -> "T5Config":
        return T5Config(d_model=128, d_ff=256, num_layers=3, num_heads=4, vocab_size=512)

    def reference_dict(self) -> dict:
        return {"d_ff": self.d_ff, "d_kv": self.d_model // self.num_heads, "d_model": self.d_model, "dense_act_fn": "gelu_pytorch_tanh",
                "is_gated_act": True, "model_type": "t5", "num_heads": self.num_heads, "num_layers": self.num_layers,
                "vocab_size": self.vocab_size}


def t5_state_dict_spec(cfg: T5Config):
    """T5's state dict in module order (FluxClip.py:386-519): encoder.block.N.layer.{0,1}.*, final norm, shared."""
    e, f = cfg.d_model, cfg.d_ff
    spec = []
    for l in range(cfg.num_layers):
        a, m = f"encoder.block.{l}.layer.0", f"encoder.block.{l}.layer.1"
        for n in ("q", "k", "v", "o"):
            spec += [(f"{a}.SelfAttention.{n}.weight", (e, e))]
        if l == 0:
            spec += [(f"{a}.SelfAttention.relative_attention_bias.weight", (cfg.num_buckets, cfg.num_heads))]
        spec += [(f"{a}.layer_norm.weight", (e,)),
                 (f"{m}.DenseReluDense.wi_0.weight", (f, e)), (f"{m}.DenseReluDense.wi_1.weight", (f, e)),
                 (f"{m}.DenseReluDense.wo.weight", (e, f)), (f"{m}.layer_norm.weight", (e,))]
    spec += [("encoder.final_layer_norm.weight", (e,)), ("shared.weight", (cfg.vocab_size, e))]
    return spec


# ------------------------------------------------------------------------------------------------------
@dataclass
class FluxConfig:
    """FluxParams (src/BlackForest/Flux.py:293-306); defaults = flux-dev."""

    in_channels: int = 16
    vec_in_dim: int = 768
    context_in_dim: int = 4096
    hidden_size: int = 3072
    mlp_ratio: float = 4.0
    num_heads: int = 24
    depth: int = 19
    depth_single_blocks: int = 38
    axes_dim: Tuple[int, ...] = (16, 56, 56)
    theta: int = 10000
    qkv_bias: bool = True
    guidance_embed: bool = True

    @property
    def mlp_hidden(self) -> int:
        return int(self.hidden_size * self.mlp_ratio)

    @staticmethod
    def tiny() -> "FluxConfig":
        """SURVEY Appendix B tiny Flux3, widened to the engine's multiples (hidden 64, 2 heads of 32)."""
        return FluxConfig(in_channels=16, vec_in_dim=32, context_in_dim=48, hidden_size=64, num_heads=2, depth=2,
                          depth_single_blocks=3, axes_dim=(8, 12, 12))

    def reference_kwargs(self) -> dict:
        return dict(in_channels=self.in_channels, vec_in_dim=self.vec_in_dim, context_in_dim=self.context_in_dim,
                    hidden_size=self.hidden_size, mlp_ratio=self.mlp_ratio, num_heads=self.num_heads, depth=self.depth,
                    depth_single_blocks=self.depth_single_blocks, axes_dim=list(self.axes_dim), theta=self.theta,
                    qkv_bias=self.qkv_bias, guidance_embed=self.guidance_embed)


def flux_state_dict_spec(cfg: FluxConfig):
    """Flux3 state-dict keys in module creation order (Flux.py:543-657)."""
    c, m, d = cfg.hidden_size, cfg.mlp_hidden, cfg.hidden_size // cfg.num_heads
    inc = 4 * cfg.in_channels
    spec = []

    def lin(name, n, k, bias=True):
        nonlocal spec
        spec += [(f"{name}.weight", (n, k))] + ([(f"{name}.bias", (n,))] if bias else [])

    lin("img_in", c, inc)
    lin("time_in.in_layer", c, 256); lin("time_in.out_layer", c, c)
    lin("vector_in.in_layer", c, cfg.vec_in_dim); lin("vector_in.out_layer", c, c)
    if cfg.guidance_embed:
        lin("guidance_in.in_layer", c, 256); lin("guidance_in.out_layer", c, c)
    lin("txt_in", c, cfg.context_in_dim)
    for i in range(cfg.depth):
        p = f"double_blocks.{i}"
        for s in ("img", "txt"):
            lin(f"{p}.{s}_mod.lin", 6 * c, c)
            lin(f"{p}.{s}_attn.qkv", 3 * c, c, cfg.qkv_bias)
            spec += [(f"{p}.{s}_attn.norm.query_norm.scale", (d,)), (f"{p}.{s}_attn.norm.key_norm.scale", (d,))]
            lin(f"{p}.{s}_attn.proj", c, c)
            lin(f"{p}.{s}_mlp.0", m, c); lin(f"{p}.{s}_mlp.2", c, m)
    for i in range(cfg.depth_single_blocks):
        p = f"single_blocks.{i}"
        lin(f"{p}.linear1", 3 * c + m, c); lin(f"{p}.linear2", c, c + m)
        spec += [(f"{p}.norm.query_norm.scale", (d,)), (f"{p}.norm.key_norm.scale", (d,))]
        lin(f"{p}.modulation.lin", 3 * c, c)
    lin("final_layer.linear", inc, c)
    lin("final_layer.adaLN_modulation.1", 2 * c, c)
    return spec
