/* ldx.h — C ABI of libldx.so, the MI355X-native drop-in for LightDiffusion-Next's denoising hot path.
 *
 * The reference (Aatricks/LightDiffusion-Next, 100 % Python) exposes one plugin boundary for an
 * accelerated UNet: model_options["model_function_wrapper"], called at src/cond/cond.py:254-263 as
 *     wrapper(model.apply_model, {"input", "timestep", "c", "cond_or_uncond"})
 * and installed with ModelPatcher.set_model_unet_function_wrapper (src/Model/ModelPatcher.py:138-144);
 * Stable-Fast (src/StableFast/StableFast.py:230-274) and FBCache (src/WaveSpeed/fbcache_nodes.py:96-111)
 * sit behind it.  Everything below is what a ctypes binding for that hook needs: plain pointers and
 * sizes, no torch types.  All `const float*` / `void*` tensor arguments of the compute calls are
 * DEVICE pointers on the engine's device; ldx_load_tensor takes HOST pointers.
 *
 * Conventions: every call returns 0 on success or a negative LDX_E* code; ldx_last_error() returns a
 * thread-local human-readable message.  One engine per device, not thread-safe (the reference has a
 * single caller thread).  `stream` is a hipStream_t passed as void* (NULL = default).
 */
#ifndef LDX_H
#define LDX_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define LDX_OK 0
#define LDX_EINVAL (-1)      /* bad argument / unsupported configuration */
#define LDX_EMISSING (-2)    /* a required weight tensor was not loaded */
#define LDX_EHIP (-3)        /* HIP runtime error */
#define LDX_ESTATE (-4)      /* call sequence error (e.g. denoise before finalize) */

/* element types */
#define LDX_BF16 0
#define LDX_F16 1
#define LDX_F32 2

typedef struct ldx_engine ldx_engine;

/* Mirrors the keyword arguments of UNetModel1.__init__ (src/NeuralNetwork/unet.py:208-252) that the
 * SD1.5 family uses (src/SD15/SD15.py:10-77 + detect_unet_config unet.py:773-1080). */
typedef struct ldx_unet_config {
    int32_t compute_dtype;              /* LDX_BF16 (default) or LDX_F16: activation + weight storage type */
    int32_t in_channels;                /* 4 */
    int32_t out_channels;               /* 4 */
    int32_t model_channels;             /* 320; must be a multiple of 64 */
    int32_t num_levels;                 /* len(channel_mult) = 4 */
    int32_t channel_mult[8];            /* 1,2,4,4 */
    int32_t num_res_blocks[8];          /* 2,2,2,2 */
    int32_t transformer_depth[32];      /* per input res block, consumed front-to-back: 1,1,1,1,1,1,0,0 */
    int32_t transformer_depth_output[48]; /* per output block, stored in the reference's order (popped from the END) */
    int32_t transformer_depth_middle;   /* 1 ; -1 = no transformer, -2 = no middle block */
    int32_t num_heads;                  /* 8 */
    int32_t context_dim;                /* 768; multiple of 64 */
} ldx_unet_config;

/* VAE decoder (AutoencoderKL) — kwargs of Decoder.__init__ (src/AutoEncoders/VariationalAE.py:416-530) with the
 * ddconfig VAE.__init__ builds (VariationalAE.py:612-640). */
typedef struct ldx_vae_config {
    int32_t compute_dtype;      /* LDX_BF16 | LDX_F16 (the reference decodes in fp32 on CPU/ROCm, VAE_DTYPE) */
    int32_t z_channels;         /* 4 */
    int32_t ch;                 /* 128 ; multiple of 64 */
    int32_t num_levels;         /* 4 */
    int32_t ch_mult[8];         /* 1,2,4,4 */
    int32_t num_res_blocks;     /* 2 */
    int32_t out_ch;             /* 3 */
    int32_t use_post_quant;     /* 1 for SD1.5 (AutoencodingEngine.decode, VariationalAE.py:130-145); 0 for Flux */
} ldx_vae_config;

/* CLIP text encoder — keys of include/clip/sd1_clip_config.json read by CLIPTextModel_ (src/clip/CLIPTextModel.py:3-50). */
typedef struct ldx_clip_config {
    int32_t compute_dtype;
    int32_t hidden_size;            /* 768 */
    int32_t num_layers;             /* 12 */
    int32_t num_heads;              /* 12 */
    int32_t intermediate_size;      /* 3072 */
    int32_t max_positions;          /* 77 */
    int32_t vocab_size;             /* 49408 */
} ldx_clip_config;

/* T5 encoder — src/clip/clip/t5_config_xxl.json as read by T5 (src/clip/FluxClip.py:476-519): gated tanh-GELU FF,
 * RMS T5LayerNorm (eps 1e-6), relative-position bias from block 0 shared by every block, inner_dim = d_model. */
typedef struct ldx_t5_config {
    int32_t compute_dtype;
    int32_t d_model;               /* 4096 ; multiple of 64 */
    int32_t d_ff;                  /* 10240 ; multiple of 64 */
    int32_t num_layers;            /* 24 */
    int32_t num_heads;             /* 64 ; head dim = d_model / num_heads, multiple of 8, <= 160 */
    int32_t vocab_size;            /* 32128 */
} ldx_t5_config;

/* ESRGAN RRDBNet (src/UltimateSDUpscale/RDRB.py:216-471): nf 64 / gc 32 dense blocks, 2^num_upscale nearest+conv upsampling. */
typedef struct ldx_esrgan_config {
    int32_t compute_dtype;
    int32_t in_nc;                 /* 3 */
    int32_t out_nc;                /* 3 */
    int32_t nf;                    /* 64 */
    int32_t gc;                    /* 32 */
    int32_t num_blocks;            /* 23 RRDBs */
    int32_t num_upscale;           /* 2 -> x4 */
} ldx_esrgan_config;

/* Flux DiT — FluxParams (src/BlackForest/Flux.py:293-306); flux-dev: 16, 768, 4096, 3072, 4.0, 24, 19, 38,
 * axes [16,56,56], theta 10000, qkv_bias 1, guidance_embed 1. */
typedef struct ldx_flux_config {
    int32_t compute_dtype;
    int32_t in_channels;          /* latent channels (16); tokens carry 4x that after the 2x2 patchify */
    int32_t vec_in_dim;           /* 768  (pooled CLIP-L) */
    int32_t context_in_dim;       /* 4096 (T5-XXL) ; multiple of 8 */
    int32_t hidden_size;          /* 3072 ; multiple of 64 */
    int32_t mlp_hidden;           /* int(hidden_size * mlp_ratio) = 12288 ; multiple of 64 */
    int32_t num_heads;            /* 24 ; head dim = hidden/heads in {16,32,64,128} */
    int32_t depth;                /* 19 double-stream blocks */
    int32_t depth_single;         /* 38 single-stream blocks */
    int32_t guidance_embed;       /* 1 */
} ldx_flux_config;

/* ---- lifecycle -------------------------------------------------------------------------------- */
const char* ldx_version(void);
const char* ldx_last_error(void);
/* Create an engine for `device` (HIP ordinal).  Replaces BaseModel.__init__ + UNetModel1.__init__
 * (src/Model/ModelBase.py:38-57, unet.py:208-677) for the accelerated path. */
int ldx_create(const ldx_unet_config* cfg, int device, ldx_engine** out);
void ldx_destroy(ldx_engine* e);
/* Hand one state-dict tensor to the engine (HOST pointer; dtype LDX_F16|LDX_BF16|LDX_F32).  Keys use
 * the SD1.5 `model.diffusion_model.` layout with that prefix stripped, e.g.
 * "input_blocks.1.1.transformer_blocks.0.attn1.to_q.weight".  Replaces BaseModel.load_model_weights
 * (ModelBase.py:178-202) + the per-forward cast of cond/cast.py:44-78 (cast happens once, here). */
int ldx_load_tensor(ldx_engine* e, const char* key, const void* data, int dtype, const int64_t* shape, int ndim);
/* Sigma / timestep tables built by the host exactly as the reference builds them
 * (ModelSamplingDiscrete.set_sigmas sampling.py:285-289 -> log_sigmas[n];
 *  timestep_embedding sampling_util.py:56-76 evaluated at t = 0..n-1 -> temb[n][model_channels]). */
int ldx_set_tables(ldx_engine* e, const float* log_sigmas, int n, const float* temb, int temb_dim);
/* Pack weights into MFMA-friendly layouts and upload them.  After this the host copies are dropped. */
int ldx_finalize(ldx_engine* e);

/* ---- the hot path ----------------------------------------------------------------------------- */
/* The wrapper body: denoised = x - UNet(x / sqrt(sigma^2+1), t(sigma), ctx) * sigma — i.e. all of
 * BaseModel.apply_model (ModelBase.py:72-133) for EPS prediction (sampling.py:26-56).
 *   x_nchw  [B2][C][h][w] fp32, sigma [B2] fp32 (sigma VALUES, as the hook passes them),
 *   ctx     [B2][M][context_dim] fp32 (c["c_crossattn"]), out_nchw like x_nchw.  */
int ldx_unet_denoise(ldx_engine* e, const float* x_nchw, const float* sigma, const float* ctx,
                     int B2, int h, int w, int M, float* out_nchw, void* stream);
/* The same with the concat conditioning of inpainting UNets (BaseModel.apply_model, ModelBase.py:100-101:
 * xc = torch.cat((x / sqrt(sigma^2 + 1), c_concat), dim=1)): x_nchw [B2][in_channels - cc_channels][h][w] is the latent (scaled, and the x of
 * denoised = x - eps * sigma), c_concat [B2][cc_channels][h][w] fp32 is appended UNSCALED; the engine's in_channels (9 for SD1.5 inpainting) counts
 * both.  LDX_EINVAL unless in_channels - cc_channels == out_channels. */
int ldx_unet_denoise_concat(ldx_engine* e, const float* x_nchw, const float* sigma, const float* ctx, const float* c_concat, int cc_channels,
                            int B2, int h, int w, int M, float* out_nchw, void* stream);
/* One CFG evaluation as calc_cond_batch builds it (cond/cond.py:186-226: input_x = cat([x] * 2), timestep = cat([sigma] * 2),
 * c_crossattn = cat([uncond, cond])): x_nchw [B][C][h][w] is read by BOTH halves of the [uncond x B ; cond x B] batch, sigma is
 * one host scalar shared by every sample, ctx [2B][M][context_dim], out_nchw [2B][C][h][w].  Same arithmetic as
 * ldx_unet_denoise on the concatenated inputs (bit-identical); replaces two device copies and a fill per sampler step. */
int ldx_unet_denoise_cfg(ldx_engine* e, const float* x_nchw, float sigma, const float* ctx,
                         int B, int h, int w, int M, float* out_nchw, void* stream);
/* The same with the timestep index supplied by the caller.  ModelSamplingDiscrete.timestep (sample/sampling.py:309-320, called from
 * BaseModel.apply_model, ModelBase.py:112) is INTEGER work: argmin_k |log(sigma) - log_sigmas[k]|.  The device lookup (ldx_unet_denoise*,
 * ldx_unet_timestep) runs the GPU's logf, which may differ from the host's log in the last bit — at a near-tie (sigma at the geometric midpoint of two
 * table entries, e.g. the `normal` scheduler's fractional timesteps) that picks the other index.  Where sigma is a host scalar (every sampler
 * loop) the host side therefore evaluates the reference's own torch expression and passes the result here: bit-exact by construction.
 * t_index in [0, n_sigmas); t_index < 0 = ldx_unet_denoise_cfg (device lookup). */
int ldx_unet_denoise_cfg_t(ldx_engine* e, const float* x_nchw, float sigma, int t_index, const float* ctx,
                           int B, int h, int w, int M, float* out_nchw, void* stream);
/* ldx_unet_denoise with per-sample timestep indices from the caller: t_index [B2] fp32 DEVICE array (integer-valued), same reasoning. */
int ldx_unet_denoise_t(ldx_engine* e, const float* x_nchw, const float* sigma, const float* t_index, const float* ctx,
                       int B2, int h, int w, int M, float* out_nchw, void* stream);
/* The device's sigma -> timestep lookup alone (the device function the boundary kernel of ldx_unet_denoise* runs): sigma [n] fp32 and
 * t_out [n] int32 are DEVICE arrays.  For tests of the index as an integer (tests/test_timestep_gpu.py). */
int ldx_unet_timestep(ldx_engine* e, const float* sigma, int n, int32_t* t_out, void* stream);
/* Context cache for a sampling run.  The reference recomputes to_k(context) / to_v(context) of all 16 cross-attentions in every step
 * (CrossAttention.forward, src/Attention/Attention.py:100-124, called from transformer.py:186-245) although `c_crossattn` is the same tensor content
 * for every step of a run (calc_cond_batch rebuilds it from the same conditioning, cond/cond.py:150-288).  enable = 1: the CALLER PROMISES that the
 * bytes behind a given ctx pointer stay unchanged until its next call of this function (any argument); the engine then converts / projects the
 * context once per (input shape, ctx pointer) and every later ldx_unet_denoise* call with that pointer reuses the projections (identical bits:
 * the same kernels produced them).  Every call of this function — also with enable = 1 again — INVALIDATES what is cached: call it after
 * rewriting a context buffer in place.  enable = 0 (the default): projections recomputed on every call, as the reference does.
 * The hook object (LdxUNetPatch) leaves it off: the reference hands the hook a fresh torch.cat every step. */
int ldx_unet_context_cache(ldx_engine* e, int enable);
/* Raw UNetModel1.forward (unet.py:679-770): x (already scaled), integer timesteps given as fp32. */
int ldx_unet_forward(ldx_engine* e, const float* x_nchw, const float* timesteps, const float* ctx,
                     int B2, int h, int w, int M, float* out_nchw, void* stream);
/* Number of kernel launches in the current plan, algorithmic FLOPs of one forward at the planned shape
 * (2*MACs of every Linear/Conv + 4*B*H*N*M*D per attention), arena bytes.  STEADY-STATE numbers: while the context cache is on
 * (ldx_unet_context_cache) the context's 16-bit copy and k|v projections are not part of them (they run once per context, not per step). */
int ldx_plan_info(ldx_engine* e, int64_t* n_launches, double* flops, int64_t* arena_bytes);
/* Shared CFG prefix.  calc_cond_batch evaluates [uncond; cond] over cat([x] * 2) and cat([sigma] * 2) (cond/cond.py:186-226): until the context enters at the
 * first cross-attention (input_blocks.1.1.transformer_blocks.0.attn2 in SD1.5) both halves of the batch hold IDENTICAL values — conv_in, the first
 * ResBlock, norm / proj_in, norm1, q|k|v, the first self-attention and its to_out compute every number twice.  ldx_unet_denoise_cfg* (the one entry
 * point that KNOWS the halves share x and sigma) plans those ops on one half and copies the results into the other half's rows where the first
 * cross-attention and the skip connections read them; the outputs equal the concatenated ldx_unet_denoise call up to the summation order of
 * GroupNorm statistics (tile shapes follow the row count).  enable = 0: every op on the full batch (bit-identical to the concatenated call);
 * 1 (default; environment LDX_CFG_SHARE overrides the default): where it pays — at least LDX_CFG_SHARE_MINROWS (8192) rows per half, below that the
 * half-batch launches no longer fill the chip (512^2 at bs = 1 measured 5.73 against 5.69 ms per step shared); 2: whenever the model has a
 * cross-attention for the prefix to end at (tests). */
int ldx_unet_cfg_share(ldx_engine* e, int enable);
/* FLOPs of the current plan as EXECUTED in steady state, and the part of ldx_plan_info's algorithmic count that the shared CFG prefix does not
 * execute (algorithmic = executed + shared).  Roofline fractions are quoted on the executed number. */
int ldx_plan_flops(ldx_engine* e, double* executed, double* shared);
/* Memory behaviour: an engine keeps the launch plan, arena and captured graph of its CURRENT input shape plus up to four earlier
 * shapes (the multi-scale samplers and HiresFix alternate between resolutions; Flux plans are per (B, h, w, prompt length)).  Each
 * cached plan holds its own arena; the cache is trimmed oldest-first to at most 4 entries and LDX_PLAN_CACHE_GIB (default 16) GiB. */
/* Per-kernel-class timing of subsequent eager forwards with HIP events recorded on the launch stream
 * (used by bench.py for the roofline line).  ldx_profile_report writes a JSON object
 * {"<kernel>": {"count", "ms", "flops", "bytes"}, ...} (algorithmic flops/bytes, summed) into buf.
 * enable = 2 keys the report by kernel class *and* op shape (tuning aid). */
int ldx_profile(ldx_engine* e, int enable, int reset);
int ldx_profile_report(ldx_engine* e, char* buf, int64_t cap);
/* Capture the planned forward into a hipGraph for replay (0 = eager launches). */
int ldx_set_graph_mode(ldx_engine* e, int enable);
/* UNet engine: how many times a forward was captured into a hipGraph and how many times a captured graph was replayed since the engine was
 * created (a captured graph is tied to the pointers of the call that recorded it: a caller that alternates buffers re-captures instead of
 * replaying — sampling.CFGDenoiser stages such inputs through one persistent buffer per shape). */
int ldx_graph_stats(ldx_engine* e, int64_t* captures, int64_t* replays);
/* The kernel-dispatch experiment switches (LDX_ATTN_PIPE, LDX_ATTN_PIPE128, LDX_ATTN_PIPE_MINWG, LDX_ATTN_PIPE_THR) are read ONCE when the library
 * is loaded — no getenv on the launch path.  Tests and same-process A/B runs that change them afterwards call this to re-read them. */
int ldx_reload_env(void);

/* ---- VAE decode and CLIP text encode (same ldx_engine handle type; load/finalize/destroy as above) ---------- */
/* Keys for ldx_load_tensor: the reference's first_stage_model state dict ("decoder.*", "post_quant_conv.*"). */
int ldx_vae_create(const ldx_vae_config* cfg, int device, ldx_engine** out);
/* VAE.decode (VariationalAE.py:690-722): z [B][z_channels][h][w] fp32 (already divided by the latent scale) ->
 * clamp((decoder(post_quant_conv(z)) + 1) / 2, 0, 1) as NHWC fp32 [B][8h][8w][3]. */
int ldx_vae_decode(ldx_engine* e, const float* z_nchw, int B, int h, int w, float* out_nhwc, void* stream);
/* VAE.encode's deterministic part (VariationalAE.py:725-760, Encoder.forward :378-413, quant_conv :160-166): pixels
 * [B][H][W][3] fp32 in [0,1] -> process_input (x*2-1) -> Encoder -> quant_conv -> moments [B][2*z_channels][H/8][W/8]
 * fp32 (mean | logvar).  Needs the "encoder.*" / "quant_conv.*" tensors to have been loaded.  The stochastic
 * DiagonalGaussianRegularizer.sample (VariationalAE.py:42-51, global CPU RNG) stays on the host. */
int ldx_vae_encode(ldx_engine* e, const float* pixels_nhwc, int B, int H, int W, float* moments_nchw, void* stream);
/* Keys: the transformer's state dict with the "text_model." prefix stripped ("embeddings.token_embedding.weight",
 * "encoder.layers.0.self_attn.q_proj.weight", ..., "final_layer_norm.weight"). */
int ldx_clip_create(const ldx_clip_config* cfg, int device, ldx_engine** out);
/* CLIPTextModel_.forward (src/clip/CLIPTextModel.py:51-107): ids [B][T] int32 -> out_last = final_layer_norm(x_L)
 * [B][T][hidden] fp32; if out_inter != NULL, out_inter = final_layer_norm(x after layer `inter_layer`)
 * (negative counts from the end, e.g. -2 = clip-skip 2; Clip.py:218-236).  Causal mask, no padding mask. */
int ldx_clip_encode(ldx_engine* e, const int32_t* ids, int B, int T, int inter_layer,
                    float* out_last, float* out_inter, void* stream);
/* Pooled output: the row of `last` [B][T][hidden] (ldx_clip_encode's out_last) at the first position whose id equals eos_token_id —
 * position 0 when there is none, as torch's argmax over an all-zero row gives (CLIPTextModel_.forward, CLIPTextModel.py:98-106) —
 * and, when the optional tensor "text_projection.weight" [hidden][hidden] was loaded, that row times its transpose
 * (CLIPTextModel.forward, CLIPTextModel.py:130,152-163; fp32).  out_pooled [B][hidden] fp32. */
int ldx_clip_pooled(ldx_engine* e, const float* last, const int32_t* ids, int B, int T, int eos_token_id,
                    float* out_pooled, void* stream);
/* Textual-inversion vectors (SDClipModel.set_up_textual_embeddings, src/SD15/SDClip.py:213-267): rows_host [n][hidden] fp32
 * (HOST pointer) become token ids vocab_size .. vocab_size + n - 1 for the following ldx_clip_encode calls; n = 0 removes
 * them.  Synchronous.  The reference rebuilds its nn.Embedding per forward; the engine keeps one side table instead. */
int ldx_clip_set_extra_embeddings(ldx_engine* e, const float* rows_host, int n);

/* ---- ESRGAN upscaler --------------------------------------------------------------------------- */
/* Keys: RRDBNet's own module names ("model.0.weight", "model.1.sub.0.RDB1.conv1.0.weight", ..., "model.1.sub.<nb>.weight",
 * "model.3.weight", "model.6.weight", "model.8.weight", "model.10.weight"); newer checkpoint layouts are renamed by the host
 * exactly as RRDBNet.new_to_old_arch does (RDRB.py:381-441). */
int ldx_esrgan_create(const ldx_esrgan_config* cfg, int device, ldx_engine** out);
/* RRDBNet.forward: pixels [B][H][W][in_nc] fp32 -> [B][sH][sW][out_nc] fp32, s = 2^num_upscale (no clamp). */
int ldx_esrgan_forward(ldx_engine* e, const float* pixels_nhwc, int B, int H, int W, float* out_nhwc, void* stream);
/* tiled_scale's feathered accumulation (src/Utilities/util.py:557-590): out/div [H][W][C] fp32 += tile [th][tw][C] * mask /
 * mask at (y0, x0); then ldx_tile_finish: out = out / div (div may be NULL), optionally clamped to [0,1] (USDU_upscaler.py:94). */
int ldx_tile_blend(const float* tile, int th, int tw, float* out, float* div, int H, int W, int C, int y0, int x0, int feather, void* stream);
int ldx_tile_finish(float* out, const float* div, int64_t n, int clamp01, void* stream);

/* First-block cache (WaveSpeed, src/WaveSpeed/first_block_cache.py:105-384 + fbcache_nodes.py:8-201; the reference's Flux
 * pipeline enables it with threshold 0.12, pipeline.py:228-231): opt-in APPROXIMATE mode — outputs differ from the exact
 * forward by design.  After double block 0, r = img_after - img_before; if mean|r_prev - r| / mean|r_prev| < threshold the
 * remaining blocks are skipped and the cached (final - after-block-0) residual of both streams is added instead; otherwise
 * they run and both caches refresh.  State resets when the shape changes or sigma[0] does not decrease between calls.
 * threshold <= 0 disables (default).  Stats count forwards that used / refreshed the cache. */
int ldx_flux_fbcache(ldx_engine* e, float residual_diff_threshold);
int ldx_flux_fbcache_stats(ldx_engine* e, int64_t* hits, int64_t* misses);
/* MX fp8 mode (BASELINE config 4 "fp8 MFMA"; opt-in, approximate, own parity class): the linears of the double / single
 * blocks run the block-scaled 16x16x128 MFMA on e4m3fn operands with one E8M0 scale per 32 consecutive k (weights
 * quantised once in ldx_finalize, activations per forward), and the batched adaLN modulation projections (img_mod / txt_mod /
 * modulation / final adaLN: one skinny GEMM) read MX weights against the MX-quantised SiLU(vec) (round 6; LDX_FLUX_MOD_FP8=0 keeps them
 * 16-bit); everything else stays 16-bit.  Call before ldx_finalize;
 * needs hidden_size % 128 == 0 and mlp_hidden % 128 == 0.  The reference runs Flux from Q8_0 weights (also 32-element
 * blocks, src/Quantize/Quantizer.py:94-112) dequantised to 16-bit, i.e. W8A16; this mode is W8A8.
 * enable = 1 (2 is an alias): the linears only; attention stays 16-bit.
 * enable = 3 (explicit opt-in; a DIFFERENT, less precise workload than mode 1): ... and, at head dim 128, QK^T and PV of the joint attention on the
 * block-scaled 32x32x64 MFMA as well (the rule of ldx_op_attention_fp8 below: Q / K quantised by the QKNorm + RoPE kernel, V by a transposing quantiser,
 * P rounded to e4m3) — the whole block is fp8 MFMA.  (Round 5 had this behind enable = 1; callers of mode 1 get the linears-only arithmetic back.) */
int ldx_flux_set_fp8(ldx_engine* e, int enable);

/* ---- T5-XXL text encoder (Flux conditioning) ---------------------------------------------------- */
/* Keys: T5's state dict ("shared.weight", "encoder.block.0.layer.0.SelfAttention.q.weight", ...,
 * "encoder.block.0.layer.1.DenseReluDense.wi_0.weight", "encoder.final_layer_norm.weight").  The relative-attention
 * embedding (encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight) stays with the host, which builds
 * the bias table exactly as T5Attention.compute_bias does (FluxClip.py:150-243). */
int ldx_t5_create(const ldx_t5_config* cfg, int device, ldx_engine** out);
/* T5.forward -> T5Stack.forward (FluxClip.py:441-519): ids [B][L] int32; bias = relative-position bias fp32
 * [num_heads][L][Lp], Lp = L rounded up to 64 (padding ignored), shared by the batch and by all blocks; no padding mask;
 * attention is unscaled (the reference pre-multiplies k by sqrt(d) to cancel SDPA's scale, :265-268).
 * out = final_layer_norm(x_L) [B][L][d_model] fp32. */
int ldx_t5_encode(ldx_engine* e, const int32_t* ids, int B, int L, const float* bias, float* out, void* stream);

/* ---- Flux DiT ------------------------------------------------------------------------------ */
/* Keys for ldx_load_tensor: Flux3's state dict ("img_in.weight", "double_blocks.0.img_mod.lin.weight", ...). */
int ldx_flux_create(const ldx_flux_config* cfg, int device, ldx_engine** out);
/* Flux3.forward (Flux.py:732-778) behind BaseModel.apply_model with CONST prediction (sampling.py:100-155):
 *   denoise != 0: out = x - Flux3(x, t = sigma, ctx, y, guidance) * sigma ; denoise == 0: raw model output.
 *   x [B][C][h][w] fp32 (h, w even), sigma [B], ctx [B][Lt][context_in_dim], y [B][vec_in_dim], guidance [B] (may be
 *   NULL when guidance_embed == 0); pe_cos / pe_sin: [Lt + h*w/4][head_dim/2] fp32 rotary tables the host builds
 *   with the reference's rope() (Flux.py:36-70) for ids = [txt_ids ; img_ids]. */
int ldx_flux_forward(ldx_engine* e, const float* x, const float* sigma, const float* ctx, const float* y,
                     const float* guidance, const float* pe_cos, const float* pe_sin,
                     int B, int h, int w, int Lt, int denoise, float* out, void* stream);

/* ---- sampler-side elementwise ops (src/sample/samplers.py, src/sample/CFG.py) ------------------ */
/* d = lerp(den_uncond, den_cond, cfg) (CFG.py:60), then
 * kind 0: Euler  x = x + ((x - d)/c0)*c1, c0 = sigma_hat, c1 = sigma_next - sigma_hat (samplers.py:308; util.py:26-37)
 * kind 1: DPM++ first order  x = c0*x - c1*d, c0 = sigma_next/sigma, c1 = expm1(-h)   (samplers.py:945-946)
 * kind 2: CFG combine only (denoised_out = d; x untouched) — low-resolution multiscale steps.
 * kind 3: noise injection  x = x + den_uncond * c0 (den_uncond = the noise tensor; samplers.py:723).
 * Same fp32 operation order as the reference expressions, no FMA contraction.  denoised_out may be NULL for kinds
 * 0, 1, 3; kind 2 REQUIRES it (LDX_EINVAL otherwise); kind 0 requires c0 != 0 (LDX_EINVAL otherwise). */
int ldx_sampler_step(int kind, float* x, const float* den_uncond, const float* den_cond, float* denoised_out,
                     int64_t n, float cfg, float c0, float c1, void* stream);
/* One pass of bislerp (src/Utilities/upscale.py:5-128; LatentUpscale for HiresFix, pipeline.py:346-366): slerp of the
 * C-vector (C <= 16) between source indices c1[i], c2[i] with ratio ratios[i] along axis 1 (width) or 0 (height);
 * fp32 NCHW in/out.  The index/ratio arrays are built by the host exactly as generate_bilinear_data does. */
int ldx_bislerp_pass(const float* in, float* out, int N, int C, int H, int W, int axis, int new_len,
                     const int32_t* c1, const int32_t* c2, const float* ratios, void* stream);
/* F.interpolate(mode="bilinear", align_corners=False) on fp32 [planes][H][W] (samplers.py:227-241). */
int ldx_bilinear(const float* in, float* out, int planes, int hin, int win, int hout, int wout, void* stream);

/* ---- single-op entry points (parity tests call the kernels through these) ----------------------- */
/* 16-bit tensors are device pointers to bf16/fp16 per `dtype`.  Layouts: see csrc/ldx_kernels.h. */
int ldx_op_convert(const float* in_f32, void* out_16, int64_t n, int dtype, int to_f32, void* stream);
int ldx_op_gemm(const void* A, int lda, const void* W, int M, int N, int K, const float* bias,
                const float* rowvec, int rowvec_ld, int rows_per_batch, int geglu,
                const void* R, int ldr, void* C, int ldc, float* Cf, int ldcf, int dtype, void* stream);
/* MX fp8 operands for the block-scaled MFMA (OCP microscaling: blocks of 32 consecutive k share one E8M0 scale
 * 2^ceil(log2(amax/448)), elements are e4m3fn = x / scale rounded to nearest even; BASELINE config 4 "fp8 MFMA").
 * ldx_op_mx_quant: 16-bit X [rows][K] (stride ldx) -> Y bytes [rows][ldy] + scales: uint32 [K/128][scales_ld], byte j of
 * word [t][r] = scale of block 4 t + j of row r.  ldx_op_gemm_mx: C = act(A W^T + bias) (+ R) on such operands (A [M][K],
 * W [N][K]); fp32 accumulation; act as ldx_kernels.h GemmArgs::act; outputs 16-bit C and / or fp32 Cf — or, with C8 / SC
 * (N % 128 == 0, no R / C / Cf), the result quantised in the epilogue exactly as ldx_op_mx_quant would quantise the 16-bit
 * C: the operand of the next block-scaled GEMM without a separate pass. */
/* LayerNorm whose output is quantised as ldx_op_mx_quant would quantise the 16-bit Y (C % 128 == 0). */
int ldx_op_layernorm_mx(const void* X, int ldx, int rows, int C, float eps, const float* gamma, const float* beta,
                        void* Y8, int ldy8, void* S8, int s8_ld, int dtype, void* stream);
/* ---- MX fp8 attention for head dim 128 (csrc/attn_mx.hip; round 5: the Flux "fp8 MFMA" mode of BASELINE config 4 with QK^T and PV on the block-scaled
 * 32x32x64 MFMA instead of 16-bit).  No reference counterpart (BlackForest/Flux.py:18-33 runs SDPA in 16 bit): pinned by the stated rule only.
 *   Q, K: ldx_op_mx_quant's format on the [rows][H * 128] matrix — e4m3 bytes, one E8M0 scale per (row, head, 32 consecutive d), scale dwords [H][s_ld]
 *         (byte j of dword [h][row] = d block j).  ldx_op_qk_norm_rope_mx produces them from the fused q|k|v projection in one pass (QKNorm + RoPE +
 *         quantiser), bit-identical to the 16-bit rope followed by ldx_op_mx_quant.
 *   V:    ldx_op_mx_vt_quant — transposed, V8T [B][H][128 d][Lp] with Lp = L rounded up to 128, one scale per (d, 32 consecutive keys) in
 *         SV [B][H][Lp / 128][128 d] dwords (byte t = keys 32 t .. 32 t + 31 of that 128-key block); inside every 64-key step the bytes are in the MFMA's
 *         contraction order: byte k holds key 32 (k >> 5) + 8 ((k & 15) >> 2) + 4 ((k >> 4) & 1) + (k & 3).
 *   P:    2^(s c - m) with m = ceil of the scaled row maximum, updated lazily (an integer exponent: P <= 4), rounded to e4m3 at the fixed scale 2^-6; the row sums add the ROUNDED values (an all-ones row of V^T).
 * Output: 16-bit O [rows][ldo] (O8 == NULL) or MX fp8 O8 / SO as ldx_op_attention_mx writes them. */
int ldx_op_qk_norm_rope_mx(const void* QKV, int ld, int rows, int L, int H, const float* qscale, const float* kscale, const float* cosT, const float* sinT, float eps,
                           void* Q8, void* K8, int ld8, void* SQ, void* SK, int s_ld, int dtype, void* stream);
int ldx_op_mx_vt_quant(const void* V, int ldv, int B, int H, int L, void* V8T, void* SV, int Lp, int dtype, void* stream);
int ldx_op_attention_fp8(const void* Q8, int ldq8, const void* SQ, int sq_ld, const void* K8, int ldk8, const void* SK, int sk_ld, const void* V8T, const void* SV, int Lp,
                         void* O, int ldo, void* O8, int ldo8, void* SO, int so_ld, int B, int H, int Nq, int Mk, float scale, int dtype, void* stream);
/* Attention (head dim 128, no mask) whose output is quantised in the epilogue exactly as ldx_op_mx_quant would quantise the
 * 16-bit O [B*Nq][H*128]: O8 bytes (row stride ldo8) + scales uint32 [H][so_ld] (one word per row and head).  Needs
 * B * H * ceil(Nq / 128) >= 16 workgroups (smaller problems: ldx_op_attention + ldx_op_mx_quant). */
int ldx_op_attention_mx(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, void* O8, int ldo8, void* SO, int so_ld,
                        int B, int H, int Nq, int Mk, float scale, int dtype, void* stream);
int ldx_op_mx_quant(const void* X, int ldx, int rows, int K, void* Y, int ldy, void* scales, int scales_ld, int dtype, void* stream);
int ldx_op_gemm_mx(const void* A8, int lda, const void* SA, int sa_ld, const void* W8, const void* SW, int sw_ld, int M, int N, int K,
                   const float* bias, int act, const void* R, int ldr, void* C, int ldc, float* Cf, int ldcf,
                   void* C8, int ldc8, void* SC, int sc_ld, int dtype, void* stream);
/* Two independent plain GEMMs C_i = A_i W_i^T + bias_i in ONE launch: the image and text token streams of a Flux double block each
 * meet their own weights (src/BlackForest/Flux.py:260-340 DoubleStreamBlock img_* / txt_* linears), and a
 * launch of its own for the 512-row text stream would leave most of the chip idle.  16-bit operands, or MX fp8 operands laid out as for
 * ldx_op_gemm_mx.  Large pairs run on the 256-row ping-pong tiles. */
int ldx_op_gemm2(const void* A1, int lda1, const void* W1, int M1, int N1, int K1, const float* bias1, void* C1, int ldc1,
                 const void* A2, int lda2, const void* W2, int M2, int N2, int K2, const float* bias2, void* C2, int ldc2, int dtype, void* stream);
int ldx_op_gemm2_mx(const void* A1, int lda1, const void* SA1, int sa_ld1, const void* W1, const void* SW1, int sw_ld1, int M1, int N1, int K1,
                    const float* bias1, void* C1, int ldc1,
                    const void* A2, int lda2, const void* SA2, int sa_ld2, const void* W2, const void* SW2, int sw_ld2, int M2, int N2, int K2,
                    const float* bias2, void* C2, int ldc2, int dtype, void* stream);
int ldx_op_conv3x3(const void* X, int ldx, const void* W, int B, int Hin, int Win, int Cin, int Cout,
                   int stride, int Hout, int Wout, int resize_to_out, const float* bias,
                   const float* rowvec, int rowvec_ld, const void* R, int ldr, void* Y, int ldy,
                   int dtype, void* stream);
/* ResBlock1 tail (ResBlock.py:315-335) as ONE implicit GEMM: Y = conv3x3(X, W[:, :9*Cin]) + X2 W[:, 9*Cin:]^T + bias, i.e. conv2(h) +
 * skip_connection(x) with W = [Cout][ky][kx][Cin | Cin2] and the two biases summed; stride 1, pad 1, X2 rows = output rows. */
int ldx_op_conv3x3_skip(const void* X, int ldx, const void* X2, int ldx2, int Cin2, const void* W, int B, int H, int Wd, int Cin, int Cout,
                        const float* bias, void* Y, int ldy, int dtype, void* stream);
int ldx_op_groupnorm(const void* X, int ldx, void* Y, int ldy, int B, int HW, int C, int G, float eps, int silu,
                     const float* gamma, const float* beta, float* workspace, int dtype, void* stream);
int64_t ldx_op_groupnorm_workspace_floats(int B, int G);
int ldx_op_layernorm(const void* X, int ldx, void* Y, int ldy, int rows, int C, float eps,
                     const float* gamma, const float* beta, int dtype, void* stream);
int ldx_op_attention(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, void* O, int ldo,
                     int B, int H, int Nq, int Mk, int D, float scale, int causal, int dtype, void* stream);
/* Cross-attention sub-block of a BasicTransformerBlock as ONE kernel (reference: transformer.py:186-245 attn2 + Attention.py:100-124):
 * H[m][:] += to_out(softmax(to_q(LayerNorm(H[m][:])) . K_b^T * scale) . V_b) + bo, in place, m in [0, M), image b = m / N.  Wq / Wo [C][C]
 * 16-bit (row = output feature), K / V = the projected context (rows b * Mk + key, head h at columns h * (C / heads)).  Shapes the kernel
 * takes: C = 320, heads = 8, Mk <= 80, N % 128 == 0, M % N == 0 — anything else returns LDX_EINVAL (use the separate ops). */
int ldx_op_xattn_block(void* H, int ldh, int64_t M, int N, int C, int heads, const float* ln_gamma, const float* ln_beta, float eps,
                       const void* Wq, const void* Wo, const float* bo, const void* K, int ldk, const void* V, int ldv, int Mk,
                       float scale, int dtype, void* stream);
/* Feed-forward sub-block of a BasicTransformerBlock as ONE kernel (reference: transformer.py:19-70, 240-244; Activation.py:6-31):
 * H[m][:] += W2 . (a * gelu_erf(g)) + b2 with [a | g] = W1 . LayerNorm(H[m][:]) + b1, in place.  W1 [2 * inner][C] and b1 in the engine's GEGLU
 * row layout: slab s (64 rows) = value rows of inner features 32 s .. 32 s + 31, then their gate rows (reference rows i and inner + i);
 * W2 [C][inner].  Shapes the kernel takes: C = 320, inner = 1280 — anything else returns LDX_EINVAL (use the separate ops). */
int ldx_op_ff_block(void* H, int ldh, int64_t M, int C, int inner, const float* ln_gamma, const float* ln_beta, float eps,
                    const void* W1, const float* b1, const void* W2, const float* b2, int dtype, void* stream);
/* Row-block GEMM with a normalisation prologue (csrc/rowgemm.hip): Y[m][0:N) = pro(X[m][0:K)) . W^T + bias (+ R[m][:]), K = 320 or 640, N a multiple of K, W [N][K].
 * pro 0: identity; 1: LayerNorm(gamma, beta, eps) (transformer.py:199 norm1 in front of to_q|k|v); 2: GroupNorm apply, 32 groups
 * (transformer.py:361-367 norm in front of proj_in) with the statistics given as partial sums partial[b][chunk][32][2] = (sum, sum of squares)
 * over any split of image b's HW pixels (HW a multiple of 128 at K = 320, of 64 at K = 640) into `nchunk` <= 256 chunks (what the producing
 * conv's epilogue writes).  Other shapes: LDX_EINVAL. */
int ldx_op_rowgemm(const void* X, int ldx, void* Y, int ldy, int64_t M, int N, int K, const void* W, const float* bias, const void* R, int ldr,
                   int pro, const float* gamma, const float* beta, float eps, const float* partial, int nchunk, int HW, int dtype, void* stream);
/* attention with an additive fp32 score bias [H][>= Nq][bias_ld] (bias_ld >= Mk rounded up to 64), added before the scale */
int ldx_op_attention_bias(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, void* O, int ldo,
                          int B, int H, int Nq, int Mk, int D, float scale, const float* bias, int bias_ld,
                          int64_t bias_head_stride, int dtype, void* stream);
int ldx_op_skinny(const float* x, int ldx, const void* W, const float* bias, float* out, int ldo,
                  int M, int N, int K, int in_act, int out_act, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LDX_H */
