"""ctypes binding of libldx.so (the C ABI declared in include/ldx.h).

There is deliberately no fallback: if the shared library is missing or a call fails, an exception is
raised.  PyTorch is used by callers only to own device memory; pointers cross this boundary as ints.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libldx.so")

LDX_BF16, LDX_F16, LDX_F32 = 0, 1, 2


class LdxError(RuntimeError):
    pass


class ldx_unet_config(C.Structure):
    _fields_ = [
        ("compute_dtype", C.c_int32),
        ("in_channels", C.c_int32),
        ("out_channels", C.c_int32),
        ("model_channels", C.c_int32),
        ("num_levels", C.c_int32),
        ("channel_mult", C.c_int32 * 8),
        ("num_res_blocks", C.c_int32 * 8),
        ("transformer_depth", C.c_int32 * 32),
        ("transformer_depth_output", C.c_int32 * 48),
        ("transformer_depth_middle", C.c_int32),
        ("num_heads", C.c_int32),
        ("context_dim", C.c_int32),
    ]


class ldx_vae_config(C.Structure):
    _fields_ = [("compute_dtype", C.c_int32), ("z_channels", C.c_int32), ("ch", C.c_int32), ("num_levels", C.c_int32),
                ("ch_mult", C.c_int32 * 8), ("num_res_blocks", C.c_int32), ("out_ch", C.c_int32), ("use_post_quant", C.c_int32)]


class ldx_clip_config(C.Structure):
    _fields_ = [("compute_dtype", C.c_int32), ("hidden_size", C.c_int32), ("num_layers", C.c_int32), ("num_heads", C.c_int32),
                ("intermediate_size", C.c_int32), ("max_positions", C.c_int32), ("vocab_size", C.c_int32)]


class ldx_esrgan_config(C.Structure):
    _fields_ = [("compute_dtype", C.c_int32), ("in_nc", C.c_int32), ("out_nc", C.c_int32), ("nf", C.c_int32), ("gc", C.c_int32),
                ("num_blocks", C.c_int32), ("num_upscale", C.c_int32)]


class ldx_t5_config(C.Structure):
    _fields_ = [("compute_dtype", C.c_int32), ("d_model", C.c_int32), ("d_ff", C.c_int32), ("num_layers", C.c_int32),
                ("num_heads", C.c_int32), ("vocab_size", C.c_int32)]


class ldx_flux_config(C.Structure):
    _fields_ = [("compute_dtype", C.c_int32), ("in_channels", C.c_int32), ("vec_in_dim", C.c_int32), ("context_in_dim", C.c_int32),
                ("hidden_size", C.c_int32), ("mlp_hidden", C.c_int32), ("num_heads", C.c_int32), ("depth", C.c_int32),
                ("depth_single", C.c_int32), ("guidance_embed", C.c_int32)]


_vp, _i, _f, _i64 = C.c_void_p, C.c_int, C.c_float, C.c_int64
_SIGS = {
    "ldx_version": (C.c_char_p, []),
    "ldx_last_error": (C.c_char_p, []),
    "ldx_create": (_i, [C.POINTER(ldx_unet_config), _i, C.POINTER(_vp)]),
    "ldx_destroy": (None, [_vp]),
    "ldx_load_tensor": (_i, [_vp, C.c_char_p, _vp, _i, C.POINTER(_i64), _i]),
    "ldx_set_tables": (_i, [_vp, _vp, _i, _vp, _i]),
    "ldx_finalize": (_i, [_vp]),
    "ldx_unet_denoise": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "ldx_unet_denoise_cfg": (_i, [_vp, _vp, C.c_float, _vp, _i, _i, _i, _i, _vp, _vp]),
    "ldx_unet_denoise_cfg_t": (_i, [_vp, _vp, C.c_float, _i, _vp, _i, _i, _i, _i, _vp, _vp]),
    "ldx_unet_denoise_t": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "ldx_unet_timestep": (_i, [_vp, _vp, _i, _vp, _vp]),
    "ldx_unet_denoise_concat": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "ldx_unet_context_cache": (_i, [_vp, _i]),
    "ldx_unet_cfg_share": (_i, [_vp, _i]),
    "ldx_plan_flops": (_i, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "ldx_reload_env": (_i, []),
    "ldx_unet_forward": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "ldx_plan_info": (_i, [_vp, C.POINTER(_i64), C.POINTER(C.c_double), C.POINTER(_i64)]),
    "ldx_profile": (_i, [_vp, _i, _i]),
    "ldx_profile_report": (_i, [_vp, C.c_char_p, _i64]),
    "ldx_set_graph_mode": (_i, [_vp, _i]),
    "ldx_graph_stats": (_i, [_vp, C.POINTER(_i64), C.POINTER(_i64)]),
    "ldx_vae_create": (_i, [C.POINTER(ldx_vae_config), _i, C.POINTER(_vp)]),
    "ldx_vae_decode": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "ldx_clip_create": (_i, [C.POINTER(ldx_clip_config), _i, C.POINTER(_vp)]),
    "ldx_clip_encode": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "ldx_clip_pooled": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "ldx_clip_set_extra_embeddings": (_i, [_vp, _vp, _i]),
    "ldx_flux_fbcache": (_i, [_vp, _f]),
    "ldx_flux_set_fp8": (_i, [_vp, _i]),
    "ldx_flux_fbcache_stats": (_i, [_vp, C.POINTER(_i64), C.POINTER(_i64)]),
    "ldx_esrgan_create": (_i, [C.POINTER(ldx_esrgan_config), _i, C.POINTER(_vp)]),
    "ldx_esrgan_forward": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "ldx_tile_blend": (_i, [_vp, _i, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "ldx_tile_finish": (_i, [_vp, _vp, _i64, _i, _vp]),
    "ldx_t5_create": (_i, [C.POINTER(ldx_t5_config), _i, C.POINTER(_vp)]),
    "ldx_t5_encode": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp]),
    "ldx_flux_create": (_i, [C.POINTER(ldx_flux_config), _i, C.POINTER(_vp)]),
    "ldx_flux_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "ldx_sampler_step": (_i, [_i, _vp, _vp, _vp, _vp, _i64, _f, _f, _f, _vp]),
    "ldx_bislerp_pass": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "ldx_vae_encode": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "ldx_bilinear": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "ldx_op_convert": (_i, [_vp, _vp, _i64, _i, _i, _vp]),
    "ldx_op_gemm": (_i, [_vp, _i, _vp, _i, _i, _i, _vp, _vp, _i, _i, _i, _vp, _i, _vp, _i, _vp, _i, _i, _vp]),
    "ldx_op_conv3x3": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _i, _vp, _i, _i, _vp]),
    "ldx_op_groupnorm": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _f, _i, _vp, _vp, _vp, _i, _vp]),
    "ldx_op_groupnorm_workspace_floats": (_i64, [_i, _i]),
    "ldx_op_conv3x3_skip": (_i, [_vp, _i, _vp, _i, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _i, _vp]),
    "ldx_op_layernorm_mx": (_i, [_vp, _i, _i, _i, _f, _vp, _vp, _vp, _i, _vp, _i, _i, _vp]),
    "ldx_op_attention_mx": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _f, _i, _vp]),
    "ldx_op_qk_norm_rope_mx": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _i, _vp, _vp, _i, _i, _vp]),
    "ldx_op_mx_vt_quant": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _i, _i, _vp]),
    "ldx_op_attention_fp8": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _f, _i, _vp]),
    "ldx_op_mx_quant": (_i, [_vp, _i, _i, _i, _vp, _i, _vp, _i, _i, _vp]),
    "ldx_op_gemm2": (_i, [_vp, _i, _vp, _i, _i, _i, _vp, _vp, _i, _vp, _i, _vp, _i, _i, _i, _vp, _vp, _i, _i, _vp]),
    "ldx_op_gemm2_mx": (_i, [_vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _i, _vp]),
    "ldx_op_gemm_mx": (_i, [_vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _vp]),
    "ldx_op_layernorm": (_i, [_vp, _i, _vp, _i, _i, _i, _f, _vp, _vp, _i, _vp]),
    "ldx_op_attention": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _f, _i, _i, _vp]),
    "ldx_op_xattn_block": (_i, [_vp, _i, _i64, _i, _i, _i, _vp, _vp, _f, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _f, _i, _vp]),
    "ldx_op_ff_block": (_i, [_vp, _i, _i64, _i, _i, _vp, _vp, _f, _vp, _vp, _vp, _vp, _i, _vp]),
    "ldx_op_rowgemm": (_i, [_vp, _i, _vp, _i, _i64, _i, _i, _vp, _vp, _vp, _i, _i, _vp, _vp, _f, _vp, _i, _i, _i, _vp]),
    "ldx_op_attention_bias": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _f, _vp, _i, _i64, _i, _vp]),
    "ldx_op_skinny": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
}
EXPORTS = tuple(_SIGS)

_lib = None


def load():
    """Load libldx.so (once).  Raises LdxError when the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LdxError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C lightdiffusion-next_amd/csrc`).  There is no CPU fallback."
            )
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = load().ldx_last_error().decode(errors="replace")
        raise LdxError(f"{what} failed with code {rc}: {msg}")


def ptr(t):
    """Device/host pointer of a torch tensor (or None)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def torch_dtype_code(dtype):
    import torch

    return {torch.bfloat16: LDX_BF16, torch.float16: LDX_F16, torch.float32: LDX_F32}[dtype]


def current_stream_ptr():
    import torch

    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
