// ldx — MI355X (gfx950 / CDNA4) kernels for the LightDiffusion-Next denoising hot path.
// Internal header: argument structs + host launchers for every device kernel.
// All activations are NHWC ("token-major") 16-bit (bf16 or fp16) with an explicit row stride
// (ld*) so that channel-concatenation (reference: torch.cat([h, hs.pop()], 1), unet.py:750)
// is a pointer offset, never a copy.  Accumulation, norm statistics and softmax are fp32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ldx {

enum DType : int { DT_BF16 = 0, DT_F16 = 1 };

// One-time hipFuncSetAttribute(MaxDynamicSharedMemorySize) PER DEVICE (function attributes are per device: a process that
// builds engines on two GPUs must raise the limit on both).  One DevOnce per launcher instantiation; bit = device ordinal.
struct DevOnce { unsigned long long mask = 0; };
inline void set_dyn_lds(DevOnce& once, const void* fn, int bytes) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (__atomic_load_n(&once.mask, __ATOMIC_RELAXED) & bit) return;
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    __atomic_fetch_or(&once.mask, bit, __ATOMIC_RELAXED);
}

// ---------------------------------------------------------------------------------------------
// GEMM / implicit-GEMM convolution:  C[M][N] = epilogue( A[M][K] * W[N][K]^T )
// mode 0: A is a plain row-major matrix (row stride lda), K % 64 == 0.
// mode 1: A is an NHWC image [B][Hin][Win][lda]; 3x3 window, pad 1, stride 1|2, optional
//         nearest-neighbour resize of the input to (Hv,Wv) fused into the gather
//         (reference Upsample1: F.interpolate(nearest) then conv, ResBlock.py:75-138).
//         K = 9*Cin with k = (ky*3+kx)*Cin + ci, Cin % 64 == 0.
// Epilogue (all optional, applied in this order): + bias[n], + rowvec[(m / rows_per_batch)][n],
//         GEGLU pairing (out = a * gelu_erf(g); reference cond/Activation.py:6-31),
//         + R[m][n] residual, store 16-bit C (ldc) and/or fp32 Cf (ldcf).
struct GemmArgs {
    const void* A;  int lda;
    const void* W;                 // [N][K] 16-bit, K contiguous
    int M, N, K;
    int mode;                      // 0 plain, 1 conv3x3
    int Cin, Hin, Win;             // conv: stored input image
    int Hv, Wv;                    // conv: virtual (resized) input extent seen by the 3x3 window
    int Hout, Wout, stride;        // conv: output extent
    int resize;                    // conv: 1 if (Hv,Wv) != (Hin,Win) -> nearest gather
    // conv: optional second input read as a 10th "tap" (1x1, same pixel as the output row): K = 9*Cin + Cin2, W = [N][9*Cin | Cin2].
    // ResBlock1's out = conv2(h) + skip_connection(x) (ResBlock.py:315-335) becomes ONE implicit GEMM: no separate 1x1 launch, no
    // write + re-read of its result.  A2 rows are the output rows ([M][lda2]); stride 1 only.
    const void* A2; int lda2; int Cin2;
    int pad0;                      // conv: 1 -> no top/left padding (VAE Downsample: F.pad (0,1,0,1) + stride-2 conv, VariationalAE.py:224-254)
    const float* bias;             // [N] or null
    const float* rowvec; int rowvec_ld; int rows_per_batch;   // per-batch channel vector or null
    int geglu;                     // 1: N is 2*inner with slab-interleaved (a|g) rows, out width N/2: a * gelu_erf(g); 2: a * gelu_tanh(g)
    int act;                       // 0 none, 1 quick-GELU x*sigmoid(1.702x) after bias (CLIP MLP, clip/Clip.py:74-77),
                                   // 2 tanh-GELU (Flux MLPs, BlackForest/Flux.py:279,388), 3 LeakyReLU(0.2) (ESRGAN, USDU_util.py:7-10)
    const float* gate; int gate_ld; // per-batch channel gate (Flux adaLN): v *= gate[(m / rows_per_batch)][n] before + R
    const void* R; int ldr;        // residual (16-bit) or null
    float oscale;                  // != 0: v *= oscale before + R   (ResidualDenseBlock_5C: x5 * 0.2 + x, RDRB.py:205)
    const void* R2; int ldr2; float oscale2;   // R2 != null: v = v * oscale2 + R2 after + R  (RRDB: out * 0.2 + x, RDRB.py:76)
    void* C; int ldc;              // 16-bit output or null
    float* Cf; int ldcf;           // fp32 output or null
    // f8 = 1 (plain mode only): A and W are MX fp8 (OCP e4m3fn bytes, K % 128 == 0) with one E8M0 scale per 32 consecutive
    // k; scales are stored K-tile-major as dwords SA[K/128][sa_ld] (byte j of dword [kt][m] = block 4*kt + j of row m)
    int f8; const uint32_t* SA; int sa_ld; const uint32_t* SW; int sw_ld;
    // C8 != null: the output is written as MX fp8 instead of 16-bit C (bias and act only; N % 32 == 0, tile width 64 or 128):
    // bytes C8[m][c8_col + n], scales SC (SA layout, row stride sc_ld) for the blocks (c8_col + n) / 32 — the operand of the
    // next block-scaled GEMM, produced without a separate quantisation pass
    void* C8; int ldc8; int c8_col; uint32_t* SC; int sc_ld;
    // ln_c1 != null: a LayerNorm over the K columns is folded into this plain 16-bit GEMM (no split-K): A = the UN-normalised rows,
    // W = W .* gamma, bias = W beta + b, ln_c1[n] = sum_k W[n][k] of the stored 16-bit values.  The kernel accumulates each row's
    // sum / sum of squares from its A fragments and stores  rstd[m] * (acc - mean[m] * ln_c1[n]) + bias[n]  (then GEGLU etc.)
    const float* ln_c1; float ln_eps;
    int splitk; float* ws;         // splitk > 1: K range split over `splitk` workgroups per tile; fp32 partials go to
                                   // ws[splitk][M][N] and a second kernel reduces them and applies the epilogue ...
    unsigned* sk_count;            // ... unless gemm_sk_fixup(a): sk_count points at zeroed per-tile counters (SK_COUNTERS of them, left at zero by every
                                   // launch) and splitk <= SK_FIXUP_MAX_S: then the partials go to private per-(tile, split) slabs — ws needs
                                   // gemm_sk_ws_floats() floats — and the LAST workgroup to finish a tile sums them in split order and runs the fused
                                   // epilogue itself (round 4: no reduce launch; gemm_common.h)
    // gn_partial != null: the consumer of C is a GroupNorm over the same [B][gn_hw][N] tensor (gn_cpg channels per group, gn_G groups): every
    // tile also writes the sum / sum of squares of the 16-bit values it stores, per group, to gn_partial[b][tile row][group][2] with
    // gn_nchunk tile rows per batch image — GroupNormArgs::partial's layout, so the GroupNorm skips its statistics pass (one full read of the
    // activation).  Only set by gemm_gn_fuse() below: no split-K / GEGLU / MX output, tile width % gn_cpg == 0, gn_hw % tile height == 0.
    float* gn_partial; int gn_cpg, gn_G, gn_hw, gn_nchunk;
    // dup_rows != 0: every 16-bit row m of C is ALSO stored at row m + dup_rows (same columns).  The shared CFG prefix (Engine::plan, share): the
    // ops in front of the first cross-attention run on one half of the [uncond; cond] batch, and what later full-batch ops read of their results
    // (skip connections, the residual stream) is written for both halves by the producer — no copy launch, no re-read.
    long dup_rows;
    int ep_general;                // experiment switch (LDX_EP_GENERAL=1, set by launch_gemm): the general output stage for every epilogue (A/B against the lean paths of gemm_common.h)
};
// Can launch_gemm(a) produce GroupNorm statistics for a consumer GroupNorm(G groups) over [B][HW][a.N]?  If yes, returns the number of
// tile rows per batch image (the consumer's chunk count) and fills a.gn_* except gn_partial; 0 = not fusable (the GroupNorm runs its own pass).
// Launch-side guard for GemmArgs::gn_partial without split-K: the tile the launcher instantiates must be the one gemm_gn_fuse() planned the partial
// layout for (chunk = one BM-row tile of an image, whole groups inside a BN-column tile); aborts otherwise instead of feeding the GroupNorm stale rows.
void gemm_gn_tile_check(const GemmArgs& a, int BM, int BN, int S);
int gemm_gn_fuse(GemmArgs& a, int HW, int G, int max_chunks);
void launch_gemm(const GemmArgs& a, DType dt, hipStream_t s);
// two independent plain GEMMs (mode 0, no split-K / GEGLU, both 16-bit or both MX) as one launch of 128x128 tiles
void launch_gemm2(const GemmArgs& a, const GemmArgs& b, DType dt, hipStream_t s);
int gemm_choose_splitk(int M, int N, int K, bool geglu);   // 1 = no split
constexpr int SK_FIXUP_MAX_S = 4;                          // in-kernel fix-up up to this many splits (the last arriver reads S slabs back to back); above: reduce launch
constexpr int SK_COUNTERS = 8192;                          // per-tile counters a caller keeps for it
#ifdef LDX_SK_FIXUP_BUILD
inline size_t gemm_sk_ws_floats(int M, int N, int S) { return (size_t)S * ((size_t)M + 255) * ((size_t)N + 255); }      // slabs of whole tiles (tile <= 256 x 256), or the [S][M][N] layout
#else
inline size_t gemm_sk_ws_floats(int M, int N, int S) { return (size_t)S * (size_t)M * (size_t)N; }      // [S][M][N] fp32 partials for the reduce launch (the slab layout only exists in fix-up builds)
#endif
bool gemm_sk_fixup(const GemmArgs& a);                     // will launch_gemm(a) reduce inside the kernel?  (opt-in: LDX_SK_FIXUP=1; measured slower than the reduce launch)
// 256-row ping-pong tiles (gemm_pp.hip; chosen by launch_gemm's cost model): bn = 128 / 160 / 256 tile width, lnf = GemmArgs::ln_c1 fold, S = K splits
void launch_gemm_pp(const GemmArgs& a, int bn, bool lnf, int S, DType dt, hipStream_t s);
void launch_gemm_pp2(const GemmArgs& a, const GemmArgs& b, int bn, DType dt, hipStream_t s);

// MX quantisation of a 16-bit [rows][K] matrix (row stride ldx): per 32-element block, scale = 2^ceil(log2(amax / 448))
// (E8M0, no clipping), y = e4m3fn(x / scale).  Y [rows][ldy] bytes; S as GemmArgs::SA (dwords [K/128][s_ld]).  K % 128 == 0.
struct MxQuantArgs { const void* X; int ldx; int rows, K; void* Y; int ldy; uint32_t* S; int s_ld; };
void launch_mx_quant(const MxQuantArgs& a, DType dt, hipStream_t s);

// Skinny GEMM for tiny M (time embedding path): out[m][n] = bias[n] + sum_k act(x[m][k]) W[n][k]
// x, out fp32; W 16-bit [N][K]; act: 0 none, 1 SiLU on the input (reference ResBlock emb_layers:
// nn.SiLU() then Linear, ResBlock.py:283-295); out_act: 1 SiLU on the output (time_embed, unet.py:334-342).
struct SkinnyArgs {
    const float* x; int ldx; const void* W; const float* bias; float* out; int ldo;
    int M, N, K; int in_act; int out_act;
    int accum;                     // 1: out += result (sums the Flux time / guidance / vector embedders)
    // W8 != null (K % 32 == 0): MX fp8 weights (e4m3 bytes [N][K], scales SW as GemmArgs::SW) and the activation MX fake-quantised while it is staged
    // (Flux fp8 mode, round 6: the 77 adaLN modulation projections stream 3.2 GB of weights per forward instead of 6.4)
    const void* W8; const uint32_t* SW; int sw_ld;
};
void launch_skinny(const SkinnyArgs& a, DType dt, hipStream_t s);

// ---------------------------------------------------------------------------------------------
// Flash attention (online softmax): O[b][n][h*D+d] = softmax(q k^T * scale) v
// q rows: Q + (b*Nq + n)*ldq + h*D ;  k/v rows: K + (b*Mk + m)*ldk + h*D  (same for V with ldv)
// D % 8 == 0, D <= 160.  causal: 1 -> key m allowed iff m <= n (CLIP text encoder).
struct AttnArgs {
    const void* Q; int ldq; const void* K; int ldk; const void* V; int ldv;
    void* O; int ldo;
    int B, H, Nq, Mk, D; float scale; int causal;
    // optional additive score bias (T5 relative-position bias): fp32 [H][>= Nq][bias_ld], bias_ld = Mk rounded up to 64,
    // shared by the batch, added to q.k BEFORE the scale (pass bias / scale); padded entries are ignored
    const float* bias; int bias_ld; long bias_hs;
    // O8 != null (only when attention_mx_out_ok(a)): the output is written as MX fp8 instead of 16-bit O — bytes
    // O8[b*Nq + n][h*D + d] (row stride ldo8) and, D being 128, one scale dword per (row, head): SO[h][b*Nq + n] (row stride so_ld)
    void* O8; int ldo8; uint32_t* SO; int so_ld;
    // optional workspace of B * H * ceil(Mk / 64) floats (attn_pipe.hip, attn_pipe128.hip): with it the pipelined kernels prove most key blocks safe from the
    // norms of their keys (Cauchy-Schwarz) instead of taking the maximum of every score; null: the exact maximum on every block
    float* knorm_ws;
    // D = 512 (attn512.hip, the VAE's single-head AttnBlock): the keys of a query block are split over `nsplit` workgroups when N / 128 query blocks
    // alone would leave CUs idle; split_ws holds attn512_ws_floats() floats of per-split partial results (un-normalised O, running maximum,
    // denominator) that a merge launch folds.  nsplit <= 1 or split_ws == null: one workgroup per query block, output written directly
    float* split_ws; int nsplit;
};
void launch_attention(const AttnArgs& a, DType dt, hipStream_t s);
int attention_dispatch_class(const AttnArgs& a);       // 3 attn512_kernel, 1 attn40p_kernel, 2 attn128p_kernel, 0 generic (attn32g / attn32 / attn)
// launch_attention's dispatch switches (LDX_ATTN_PIPE, LDX_ATTN_PIPE128, LDX_ATTN_PIPE_MINWG, LDX_ATTN_PIPE_THR) are read once at load;
// this re-reads them (tests / same-process A/B runs only — never on the launch path)
void reload_dispatch_env();

// Software-pipelined D = 40 kernel (attn_pipe.hip, round 4): attn_pipe_ok() says whether it takes the shape (D = 40, Nq % 256 == 0, Mk % 128 == 0,
// Mk >= 256, no mask / bias); thr_override = NaN keeps the type's rescale threshold (tests force the rare path with small values).
bool attn_pipe_ok(const AttnArgs& a);
void launch_attn_pipe(const AttnArgs& a, DType dt, hipStream_t s, float thr_override);
// The same pipeline for D = 128 (attn_pipe128.hip: Flux joint attention), 16-bit or MX fp8 output (AttnArgs::O8)
bool attn_pipe128_ok(const AttnArgs& a);
// Flash attention for D = 512 heads (attn512.hip): attn512_ok() says whether launch_attention takes the shape; the planner asks attn512_splits() for the
// key split and provides attn512_ws_floats() floats of workspace in AttnArgs::split_ws
bool attn512_ok(const AttnArgs& a);
int attn512_splits(const AttnArgs& a);
size_t attn512_ws_floats(const AttnArgs& a, int nsplit);
void launch_attn512(const AttnArgs& a, DType dt, hipStream_t s);
void launch_attn_knorm(const AttnArgs& a, DType dt, hipStream_t s);      // key-block norms for either pipelined kernel (AttnArgs::knorm_ws)
void launch_attn_pipe128(const AttnArgs& a, DType dt, hipStream_t s, float thr_override);

// Cross-attention sub-block as one kernel (xattn_block.hip): H[m][:] += to_out(softmax(to_q(LayerNorm(H[m][:])) . K_b^T) . V_b) + bo, in place,
// for m in [0, M), image b = m / N.  Wq / Wo: [C][C] 16-bit, row = output feature.  K / V: the projected context, rows b * Mk + key, head h at
// columns h * D.  xattn_block_ok() says whether the kernel takes the shape (C = 320, 8 heads, Mk <= 80, N % 128 == 0; LDX_XATTN_FUSE=0: never).
struct XAttnArgs {
    void* H; int ldh; long M; int N, C, heads;
    const float* ln_g; const float* ln_b; float eps;
    const void* Wq; const void* Wo; const float* bo;
    const void* K; int ldk; const void* V; int ldv; int Mk;
    float scale;
};
// Feed-forward sub-block as one kernel (ff_block.hip): H[m][:] += W2 . GEGLU(W1 . LayerNorm(H[m][:]) + b1) + b2, in place.  W1 [2 * inner][C] in
// the engine's GEGLU row layout (slabs of 32 value rows + their 32 gate rows), erf GELU; W2 [C][inner].  ff_block_ok(): C = 320, inner = 1280.
struct FFBlockArgs {
    void* H; int ldh; long M; int C, inner;
    const float* ln_g; const float* ln_b; float eps;
    const void* W1; const float* b1; const void* W2; const float* b2;
};
bool ff_block_ok(const FFBlockArgs& a);
void launch_ff_block(const FFBlockArgs& a, DType dt, hipStream_t s);
// Row-block GEMM with a normalisation prologue (rowgemm.hip): Y[m][0:N) = pro(X[m][0:K)) . W^T + bias (+ R), K = 320 or 640, N a multiple of K.
// pro 0: identity, 1: LayerNorm(g, b, eps), 2: GroupNorm apply (32 groups; statistics = the producer-written partial sums of the consumer
// GroupNorm's layout, `nchunk` rows per image of HW pixels).  Y may be X's own buffer only for pro = 0 with R = Y (rows are workgroup-private).
struct RowGemmArgs {
    const void* X; int ldx; void* Y; int ldy; long M; int N, K;
    const void* W; const float* bias; const void* R; int ldr;
    int pro; const float* g; const float* b; float eps;
    const float* partial; int nchunk, HW, G;
    // gn_out != null (engine only, pro != 2, N == K): the consumer of Y is a GroupNorm over [M / HW][HW][N] — every workgroup also writes the sum / sum of
    // squares of the 16-bit values it stores, per group, to gn_out[b][chunk = its row block within the image][32][2] (gn_nchunk = HW / rows per block)
    float* gn_out; int gn_nchunk;
    int abl;                       // timing ablations (LDX_RG_ABL, set by the launcher; wrong results): 1 no residual loads, 2 no MFMA loop, 4 no row loads, 8 no stores
    long dup_rows;                 // != 0: row m of Y is also stored at row m + dup_rows (GemmArgs::dup_rows)
};
bool rowgemm_ok(const RowGemmArgs& a);
void launch_rowgemm(const RowGemmArgs& a, DType dt, hipStream_t s);
bool xattn_block_ok(const XAttnArgs& a);
void launch_xattn_block(const XAttnArgs& a, DType dt, hipStream_t s);
bool attention_mx_out_ok(const AttnArgs& a);      // true if launch_attention will take a kernel that implements O8 / SO

// ---------------------------------------------------------------------------------------------
// GroupNorm(32 groups) over NHWC + optional SiLU.  Two launches: partial statistics, then apply.
struct GroupNormArgs {
    const void* X; int ldx; void* Y; int ldy;
    int B, HW, C, G; float eps; int silu;
    const float* gamma; const float* beta;
    float* partial;                // workspace [B][GN_NCHUNK][G][2]
    int nchunk;                    // pixel chunks actually used (<= GN_NCHUNK), set by the launcher
    int stats_chunks;              // > 0: the producer's epilogue already wrote `stats_chunks` partial rows per batch (GemmArgs::gn_partial): no statistics
                                   // pass; more than GN_NCHUNK rows are first folded (fixed order) into partial[B][GN_FOLD][G][2] behind them
};
constexpr int GN_NCHUNK = 256;
constexpr int GN_FOLD = 64;         // fold target: partial rows per batch after folding a long producer list
constexpr int GN_MAX_PRODUCER_CHUNKS = 32768;      // most partial rows per batch a producer may write (VAE 2048^2: 4 Mi pixels in 128-row tiles)
// floats of GroupNorm workspace for B images whose largest GroupNorm'ed map has HWmax pixels: producer rows (>= GN_NCHUNK) + the fold target
static inline size_t gn_workspace_rows(long HWmax) { long r = HWmax / 64; if (r < GN_NCHUNK) r = GN_NCHUNK; if (r > GN_MAX_PRODUCER_CHUNKS) r = GN_MAX_PRODUCER_CHUNKS; return (size_t)r; }
void launch_groupnorm(const GroupNormArgs& a, DType dt, hipStream_t s);
// the dispatch rule of launch_groupnorm for its one-launch kernel (statistics + apply in one workgroup per (image, group)); the planner asks it too:
// such a GroupNorm does not take producer-written statistics (LDX_GN_SMALL_MAX overrides the element bound)
bool gn_uses_small_kernel(int B, long HW, int C, int G);

// LayerNorm over the last dim C of [rows][ldx] -> [rows][ldy]
// gamma/beta may be null (elementwise_affine=False); optional adaLN modulation (Flux):
// y = (1 + scale[b][c]) * y + shift[b][c] with b = row / rows_per_batch.  C <= 3072.
struct LayerNormArgs {
    const void* X; int ldx; void* Y; int ldy; int rows, C; float eps;
    const float* gamma; const float* beta;
    const float* scale; const float* shift; int mod_ld; int rows_per_batch;
    int rms;                       // 1: T5LayerNorm (no mean subtraction, no beta): x * rsqrt(mean(x^2) + eps) * gamma
    // Y8 != null (C % 32 == 0): the output is written as MX fp8 instead of 16-bit Y: bytes Y8[row][c] (row stride ldy8) and
    // scales S8 in the GemmArgs::SA layout (row stride s8_ld), exactly what launch_mx_quant would produce from Y
    void* Y8; int ldy8; uint32_t* S8; int s8_ld;
};
void launch_layernorm(const LayerNormArgs& a, DType dt, hipStream_t s);

// ---------------------------------------------------------------------------------------------
// Edge / scheduler kernels (fp32 NCHW at the boundary, reference ModelBase.py:72-133).
// prep: xc = x / sqrt(sigma^2 + 1) -> NHWC 16-bit with Cpad channels (zeros above C);
//       t = argmin_k |ln sigma - log_sigmas[k]| ; temb_out[b][:] = temb_table[t][:]
struct PrepArgs {
    const float* x; const float* sigma; int B, C, H, W; int Cpad; void* xc;     // xc [B][H*W][Cpad]
    const float* log_sigmas; int n_sigmas; const float* temb_table; int temb_dim;
    float* temb_out; float* t_out;   // [B][temb_dim], [B]
    int scale_input;                 // 1: divide by sqrt(sigma^2+1) ; 0: raw (ldx_unet_forward)
    const float* t_in;               // if non-null: timestep indices given directly (no sigma lookup)
    int xB;                          // batch of x (0 = B): sample b reads x[b % xB] — the [uncond; cond] halves of a CFG evaluation share one latent
    // c_concat (ModelBase.py:100-101: xc = cat((x / sqrt(sigma^2 + 1), c_concat), 1); inpainting UNets, in_channels = 9): x carries Cx channels,
    // channels Cx .. C - 1 come UNSCALED from cc [B][C - Cx][H][W].  cc == null: x carries all C channels (Cx is ignored)
    const float* cc; int Cx;
    // emb_table != null: out_emb[b][0:emb_n) = emb_table[t][0:emb_n) — the row of the per-timestep table of every ResBlock's emb_layers output
    // (Engine::build_emb_table): the time-embedding MLP is a pure function of the integer timestep, so it is evaluated once per timestep at load
    const float* emb_table; int emb_n; float* emb_out;
};
void launch_prep(const PrepArgs& a, DType dt, hipStream_t s);
// out[i] = nearest log-sigma table index of sigma[i] (the prep kernel's own lookup; ldx_unet_timestep)
void launch_timestep(const float* sigma, const float* log_sigmas, int n_sigmas, int n, int* out, hipStream_t s);
// finish: out_nchw[b][c][p] = x_nchw[b][c][p] - eps_nhwc[b][p][c] * sigma[b]   (or raw eps if x == null)
struct FinishArgs { const float* eps; int ld; const float* x; const float* sigma; float* out; int B, C, HW; int xB; };       // xB as in PrepArgs
void launch_fill_f32(float* dst, float v, int n, hipStream_t s);
void launch_fill2_f32(float* a, float va, float* b, float vb, int n, hipStream_t s);      // a[i] = va, b[i] = vb (the sigma and timestep-index slots of a CFG evaluation: one launch)
// dst[r][0:C) = src[r][0:C) for r in [0, rows): 16-bit elements, both with row stride ld, C % 8 == 0 (the shared CFG prefix's hand-over, Engine::op_dup)
void launch_dup_rows(const void* src, void* dst, int rows, int C, int ld, DType dt, hipStream_t s);
// CLIP pooled output: row of last[b] at the first position whose id == eos_id (position 0 if none: torch argmax of an all-zero row),
// then, if proj != null, out[b] = row @ proj^T (proj [E][E] fp32, row-major [out][in])
void launch_clip_pooled(const float* last, const int* ids, int B, int T, int E, int eos_id, const float* proj, float* out, hipStream_t s);
void launch_finish(const FinishArgs& a, hipStream_t s);

// 16-bit <-> fp32 conversion helpers for tests / host plumbing
void launch_f32_to_t(const float* in, void* out, size_t n, DType dt, hipStream_t s);
void launch_t_to_f32(const void* in, float* out, size_t n, DType dt, hipStream_t s);
// NCHW fp32 <-> NHWC 16-bit (VAE boundary)
void launch_nchw_to_nhwc(const float* in, void* out, int B, int C, int HW, int Cpad, float scale, DType dt, hipStream_t s);

// VAE boundary (AutoEncoders/VariationalAE.py:130-145, 690-722): z fp32 NCHW [B][C][HW] -> optional 1x1 mix
// (post_quant_conv, fp32: out[c] = b[c] + sum_k w[c][k] z[k]) -> NHWC 16-bit with Cpad channels.
void launch_vae_prep(const float* z, void* out, int B, int C, int HW, int Cpad, const float* mix_w, const float* mix_b, DType dt, hipStream_t s);
// encoder input: pixels NHWC fp32 [B][HW][C] in [0,1] -> x*2-1 (process_input, VariationalAE.py:593) -> NHWC 16-bit, Cpad channels
void launch_pixels_prep(const float* px, void* out, int B, int C, int HW, int Cpad, float scale, float shift, DType dt, hipStream_t s);
// pixel post-process: out = clamp((x + 1) / 2, 0, 1) on fp32 (process_output, VariationalAE.py:595-597)
void launch_clamp01(const float* in, float* out, size_t n, hipStream_t s);
// row softmax in place on a 16-bit [rows][ld] matrix: p = softmax(x * scale) (VAE AttnBlock, D = 512 single head)
void launch_softmax_rows(void* X, int rows, int cols, int ld, float scale, DType dt, hipStream_t s);
// CLIP embeddings (clip/Clip.py:254-294): x[b][t][:] = tok[id[b][t]][:] + pos[t][:]  (fp32 tables -> 16-bit)
// ids in [vocab, vocab + n_extra) read row id - vocab of `extra` (textual-inversion vectors, SD15/SDClip.py:213-267)
void launch_clip_embed(const int* ids, const float* tok, const float* pos, void* out, int B, int T, int C, int vocab, const float* extra, int n_extra, DType dt, hipStream_t s);

// First-block cache helpers on the joint token buffer X [B][L][C] (16-bit; rows [0, Lt) text, [Lt, L) image per batch).
// fb_diff: sums[0] = sum |(X - S0) - F| , sums[1] = sum |F| over the image rows (deterministic two-stage reduction through
// `partial`, >= 2 * 1024 floats).  fb_first: F = X - S0 (image rows, fp32).  fb_residual: R = X - S1 (all rows, fp32).
// fb_apply: X += R.
void launch_fb_diff(const void* X, const void* S0, const float* F, int B, int L, int Lt, int C, float* partial, float* sums, DType dt, hipStream_t s);
void launch_fb_first(const void* X, const void* S0, float* F, int B, int L, int Lt, int C, DType dt, hipStream_t s);
void launch_fb_residual(const void* X, const void* S1, float* R, size_t n, DType dt, hipStream_t s);
void launch_fb_apply(void* X, const float* R, size_t n, DType dt, hipStream_t s);

// tiled_scale blending (Utilities/util.py:406-600): out[y0+y][x0+x][c] += tile[y][x][c] * mask(y, x), div += mask, where mask
// ramps (t + 1) / feather over the first / last `feather` rows and columns of the tile (all four edges, skipped per axis if
// feather >= tile extent); finish: out = clamp(out / div, 0, 1) (USDU_upscaler.py:94).  NHWC fp32, C channels.
void launch_tile_blend(const float* tile, int th, int tw, float* out, float* div, int H, int W, int C, int y0, int x0, int feather, hipStream_t s);
void launch_tile_finish(float* out, const float* div, size_t n, int clamp01, hipStream_t s);

// Flux: per-head RMSNorm of q and k (QKNorm, BlackForest/Flux.py:148-200, eps 1e-6) followed by RoPE
// (apply_rope :73-82) in place on a fused [rows][ld] q|k|v buffer (q at column 0, k at column C = H*D).
// cos/sin: [L][D/2] fp32 tables built by the host exactly as rope() does (:36-70); token = row % L.
struct QkRopeArgs {
    void* QKV; int ld; int rows; int L; int H, D; const float* qscale; const float* kscale;
    const float* cosT; const float* sinT; float eps;
    // launch_qk_norm_rope_mx only (D = 128): the normalised + rotated q / k leave as MX fp8 in ldx_op_mx_quant's format instead of being written back in 16 bit:
    // Q8 / K8 [row8 + row][ld8] bytes (head h at columns h * 128), SQ / SK dwords [H][s8_ld] (byte j of dword [h][row8 + row] = the scale of d block j)
    void* Q8; void* K8; int ld8; uint32_t* SQ; uint32_t* SK; int s8_ld; long row8;
};
void launch_qk_norm_rope(const QkRopeArgs& a, DType dt, hipStream_t s);
void launch_qk_norm_rope_mx(const QkRopeArgs& a, DType dt, hipStream_t s);
// MX fp8 attention for D = 128 (attn_mx.hip).  Q8 / K8 + SQ / SK: the format above, rows b * Nq + q / b * Mk + key.  V8T [B][H][128][Lp] + SV [B][H][Lp / 128][128]:
// launch_mx_vt_quant's output (V transposed, keys in the MFMA's order inside every 64-key step, one scale per (d, 32 consecutive keys)); Lp = Mk rounded up to 128.
// Output: 16-bit O [rows][ldo] (head h at columns h * 128), or MX fp8 O8 / SO exactly as AttnArgs::O8 / SO.
struct AttnMxArgs {
    const void* Q8; int ldq8; const uint32_t* SQ; int sq_ld;
    const void* K8; int ldk8; const uint32_t* SK; int sk_ld;
    const void* V8T; const uint32_t* SV; int Lp;
    void* O; int ldo;
    void* O8; int ldo8; uint32_t* SO; int so_ld;
    int B, H, Nq, Mk; float scale;
};
bool attn_mx_ok(const AttnMxArgs& a);
void launch_attn_mx(const AttnMxArgs& a, DType dt, hipStream_t s);
// V (16 bit, rows b * L + token, head h at columns h * 128 of a row of ldv elements) -> V8T / SV (see AttnMxArgs)
struct MxVtArgs { const void* V; int ldv; int B, H, L; void* V8T; uint32_t* SV; int Lp; };
void launch_mx_vt_quant(const MxVtArgs& a, DType dt, hipStream_t s);
// timestep_embedding_flux (sample/sampling_util.py:78-104): out[b][:] = [cos(1000 t f_j) | sin(1000 t f_j)], dim 256
void launch_flux_temb(const float* t, float* out, int B, int dim, float factor, hipStream_t s);
void launch_silu_f32(const float* in, float* out, size_t n, hipStream_t s);
// patchify  x[B][C][H][W] fp32 -> tokens [B*(H/2)*(W/2)][4C] 16-bit, column = c*4 + ph*2 + pw   (Flux3.forward :742-748)
void launch_flux_patchify(const float* x, void* out, int B, int C, int H, int W, DType dt, hipStream_t s);
// unpatchify + CONST.calculate_denoised (sampling.py:108-122): out = x - tok*sigma (or tok if x == null), fp32 NCHW
void launch_flux_unpatchify(const float* tok, int ld, const float* x, const float* sigma, float* out, int B, int C, int H, int W, hipStream_t s);

// Sampler elementwise kernels (fp32, reference samplers.py / CFG.py):
//  d = lerp(den_uncond, den_cond, cfg)                             (torch.lerp, CFG.py:60)
//  kind 0 euler: x = x + ((x - d) / c0) * c1   c0 = sigma_hat, c1 = sigma_next - sigma_hat (samplers.py:308, util.py:26-37)
//  kind 1 dpmpp: x = c0 * x - c1 * d           c0 = sigma_next/sigma, c1 = expm1(-h)       (samplers.py:945-946)
//  kind 2      : denoised_out = d only (multiscale steps combine at low resolution first)
//  kind 3      : x = x + den_uncond * c0   (ancestral noise injection: x + noise * s_noise * sigma_up, samplers.py:728-729)
// den_*: the two [B] chunks [uncond; cond] returned by the wrapper (cond.py:194-195 order).
struct StepArgs {
    float* x; const float* den_uncond; const float* den_cond; float* denoised_out;  // denoised_out optional
    size_t n; float cfg; int kind;
    float c0, c1;
};
void launch_sampler_step(const StepArgs& a, hipStream_t s);
// one pass of bislerp (Utilities/upscale.py:5-128): per-pixel spherical interpolation of the C-vector between source
// positions c1[i], c2[i] with ratio r[i] along the last (axis = 1) or second-to-last (axis = 0) spatial axis.
void launch_bislerp_pass(const float* in, float* out, int N, int C, int H, int W, int axis, int new_len,
                         const int* c1, const int* c2, const float* r, hipStream_t s);
// VAE encoder tail: moments_nchw[b][c][p] = bias[c] + sum_k w[c][k] * in_nhwc[b][p][k]   (quant_conv 1x1, fp32)
void launch_mix_nhwc_to_nchw(const float* in, int ld, float* out, int B, int C, int HW, const float* w, const float* bias, hipStream_t s);
// bilinear resize (align_corners=False, antialias=False) of fp32 NCHW planes
void launch_bilinear(const float* in, float* out, int planes, int Hin, int Win, int Hout, int Wout, hipStream_t s);

}  // namespace ldx
