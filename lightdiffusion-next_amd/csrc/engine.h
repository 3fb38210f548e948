// Internal declarations of the ldx UNet engine (see engine.cpp).
#pragma once
#include <functional>
#include <initializer_list>
#include <map>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../../include/ldx.h"
#include "ldx_kernels.h"

namespace ldx {

extern thread_local std::string g_last_error;
void set_error(const std::string& s);

struct HostTensor {
    int dtype = LDX_F32;
    std::vector<int64_t> shape;
    std::vector<uint8_t> data;
    size_t numel = 0;
    float at(size_t i) const;
};

struct LinearW { void* w = nullptr; float* b = nullptr; int N = 0, K = 0;
                 void* w8 = nullptr; uint32_t* sw = nullptr; };     // MX fp8 copy + E8M0 scales [K/128][N] (Flux fp8 mode)
struct NormW { float* g = nullptr; float* b = nullptr; int C = 0; };
struct ResW { NormW gn1, gn2; LinearW conv1, conv2, skip; bool has_skip = false; bool fused_skip = false; /* fused_skip: conv2 holds [W2 | Wskip], bias b2 + bskip */ int Cin = 0, Cout = 0; int emb_off = 0;
              float eps = 1e-5f; bool has_emb = true; };
struct XfBlockW { NormW ln1, ln2, ln3; LinearW qkv, o1, q2, kv2, o2, ff1, ff2; LinearW qkv_f, q2_f, ff1_f; /* LayerNorm-folded copies (ln_fold) */ int kv_off = 0;
    // ln_fold: folded copies exist: norm1/2/3 folded into qkv_f / q2_f / ff1_f (weights W .* gamma, bias W beta + b, c1_* = row sums of the stored weights);
    // the planner picks them per input shape (small row counts, where the row-block kernels are not taken):
    // the GEMM reads the un-normalised rows and applies rstd * (acc - mean * c1) + bias in its epilogue (GemmArgs::ln_stat)
    bool ln_fold = false; float *c1_qkv = nullptr, *c1_q2 = nullptr, *c1_ff1 = nullptr; };
struct XfW { NormW gn; LinearW proj_in, proj_out; std::vector<XfBlockW> blocks; int C = 0, depth = 0; };
struct BlockW { bool has_res = false, has_xf = false, has_down = false, has_up = false; ResW res; XfW xf; LinearW down, up; int skip_ch = 0; };

// activation view inside the arena: rows x C 16-bit elements, row stride ld, starting at column col
struct Act { bool valid = false; bool owned = false; size_t off = 0; int rows = 0, C = 0, ld = 0, col = 0; };

enum OpKind { OP_PREP, OP_CVT, OP_SKINNY, OP_GEMM, OP_GN, OP_LN, OP_ATTN, OP_FINISH,
              OP_VAEPREP, OP_SOFTMAX, OP_CLAMP, OP_EMBED, OP_CVT_OUT,
              OP_PIXPREP, OP_MOMENTS, OP_COPY_OUT,
              OP_FX_PATCH, OP_FX_TEMB, OP_FX_SILU, OP_FX_ROPE, OP_FX_UNPATCH, OP_FX_CVT_CTX, OP_FX_SKINNY_Y, OP_FX_SKINNY_G, OP_MXQ, OP_GEMM2, OP_XATTN, OP_FFBLOCK, OP_ROWGEMM, OP_ATTN_MX, OP_MXVT, OP_DUP };
enum EngineKind { KIND_UNET = 0, KIND_VAE = 1, KIND_CLIP = 2, KIND_FLUX = 3, KIND_T5 = 4, KIND_ESRGAN = 5 };
struct Op {
    OpKind kind; const char* name;
    GemmArgs g; GemmArgs g2; GroupNormArgs gn; LayerNormArgs ln; AttnArgs at; SkinnyArgs sk; QkRopeArgs rp; MxQuantArgs mq; XAttnArgs xa; FFBlockArgs fb; RowGemmArgs rg; AttnMxArgs am; MxVtArgs vt;
    void* cvt_out; size_t cvt_n;
    bool ctx_only = false;         // depends on the context alone (16-bit copy of ctx, the batched k|v projection): skipped while Engine::ctx_cache holds
    // generic slots for the small ops: src/dst pointers + dims
    const void* p0; void* p1; int i0, i1, i2, i3; float f0, f1;
    double flops; double bytes; char klabel[48];
};
struct ProfEntry { long count = 0; double ms = 0, flops = 0, bytes = 0; };

struct EmbSrc { const HostTensor* w; const HostTensor* b; int n; };

struct VaeAttnW { NormW norm; LinearW q, k, v, proj; LinearW qkv; /* C = 512: fused q | k | v projection [3C][C] for the flash kernel (attn512.hip); bias = [bq | bk | 0], bv folded into proj's */ };
struct ClipLayerW { NormW ln1, ln2; LinearW qkv, out, fc1, fc2; };
struct RdbW { LinearW c[5]; };
struct T5LayerW { NormW ln1, ln2; LinearW qkv, o, wi, wo; };
struct FluxStreamW { LinearW qkv, proj, mlp0, mlp2; float* qs = nullptr; float* ks = nullptr; int mod_off = 0; };
struct FluxDoubleW { FluxStreamW img, txt; };
struct FluxSingleW { LinearW lin1_qkv, lin1_mlp, lin2; float* qs = nullptr; float* ks = nullptr; int mod_off = 0; };

class Engine {
public:
    Engine(const ldx_unet_config& c, int dev);
    Engine(const ldx_vae_config& c, int dev);
    Engine(const ldx_clip_config& c, int dev);
    Engine(const ldx_flux_config& c, int dev);
    Engine(const ldx_t5_config& c, int dev);
    Engine(const ldx_esrgan_config& c, int dev);
    ldx_esrgan_config ecfg{};
    std::vector<RdbW> es_rdb; LinearW es_first, es_trunk, es_hr, es_last; std::vector<LinearW> es_up;
    int finalize_esrgan();
    int plan_esrgan(int B, int H, int W);
    int run_esrgan(const float* px, int B, int H, int W, float* out, hipStream_t st);
    ldx_t5_config tcfg{};
    std::vector<T5LayerW> t5_layers; NormW t5_final_ln; float* t5_tok = nullptr; const float* b_bias = nullptr;
    int finalize_t5();
    int plan_t5(int B, int L);
    int run_t5(const int* ids, int B, int L, const float* bias, float* out, hipStream_t st);
    ldx_flux_config fcfg{};
    int finalize_flux();
    // First-block cache (WaveSpeed/first_block_cache.py:105-384, fbcache_nodes.py:8-201): opt-in approximate mode
    float fb_threshold = 0.f; bool fb_have_first = false, fb_have_res = false; float fb_prev_t = 0.f; bool fb_prev_valid = false;
    size_t fb_a_end = 0, fb_b_end = 0;                 // op ranges: [0, a_end) through double block 0, [a_end, b_end) the rest
    void *fb_s0 = nullptr, *fb_s1 = nullptr; float *fb_first = nullptr, *fb_res = nullptr, *fb_part = nullptr;
    void* fb_x = nullptr; int fb_B = 0, fb_L = 0, fb_Lt = 0, fb_C = 0;
    long fb_hits = 0, fb_misses = 0;
    void fb_reset() { fb_have_first = fb_have_res = false; fb_prev_valid = false; }
    // MX fp8 mode (BASELINE config 4 "fp8 MFMA"): the block linears run on block-scaled fp8 operands; opt-in, own parity class
    bool fx_fp8 = false;
    bool fx_fp8_attn = false;        // ... and QK^T / PV of the joint attention on MX fp8 too (attn_mx.hip; ldx_flux_set_fp8 mode 1, head dim 128)
    bool mx_quantize_weight(LinearW& w);
    int plan_flux(int B, int h, int w, int Lt);
    int run_flux(const float* x, const float* sigma, const float* ctx, const float* y, const float* guidance,
                 const float* pe_cos, const float* pe_sin, int B, int h, int w, int Lt, bool denoise, float* out, hipStream_t st);
    EngineKind kind = KIND_UNET;
    ldx_vae_config vcfg{};
    ldx_clip_config ccfg{};
    int finalize_vae();
    int finalize_clip();
    int plan_vae(int B, int h, int w);
    int plan_clip(int B, int T, int inter);
    int run_vae(const float* z, int B, int h, int w, float* out_nhwc, hipStream_t st);
    int plan_vae_encode(int B, int H, int W);
    int run_vae_encode(const float* px, int B, int H, int W, float* moments, hipStream_t st);
    int run_clip(const int* ids, int B, int T, int inter_layer, float* out_last, float* out_inter, hipStream_t st);
    int set_clip_extra(const float* rows_host, int n);
    ~Engine();
    int validate() const;
    int load_tensor(const char* key, const void* data, int dtype, const int64_t* shape, int ndim);
    int set_tables(const float* ls, int n, const float* temb, int dim);
    int finalize();
    // c_concat [B2][cc_channels][h][w] (fp32, may be null): appended unscaled behind the scaled x, which then carries in_channels - cc_channels channels
    // t_idx [B2] (device fp32, may be null; denoise only): timestep indices supplied by the caller instead of the device's own sigma -> index lookup
    int run(const float* x, const float* sigma_or_t, const float* ctx, int B2, int h, int w, int Mc, float* out, bool denoise, hipStream_t st, int xB = 0,
            const float* c_concat = nullptr, int cc_channels = 0, const float* t_idx = nullptr);
    int clip_pooled(const float* last, const int* ids, int B, int T, int eos_id, float* out, hipStream_t st);
    // one CFG evaluation: x [B] is read by both halves of the [uncond; cond] batch (cond.py:186-226), sigma is one host scalar for every sample
    // t_index >= 0: the sigma -> timestep index computed by the caller (the reference's own host arithmetic); < 0: the device lookup
    int run_cfg(const float* x, float sigma, const float* ctx, int B, int h, int w, int Mc, float* out, hipStream_t st, int t_index = -1);
    int timestep_lookup(const float* sigma_dev, int n, int* out_dev, hipStream_t st);      // ldx_unet_timestep: the prep kernel's lookup alone
    float* d_sigma_cfg = nullptr; int sigma_cfg_cap = 0;      // [2 * cap] sigma followed by [2 * cap] timestep indices
    // ---- step-invariant work (round 5) ----
    // (1) per-timestep table of the 22 emb_layers outputs: time_embed -> SiLU -> emb_layers is a pure function of the INTEGER timestep (a8: t = argmin
    //     index; unet.py:333-342, ResBlock.py:283-295), so all n_sigmas rows are computed once by the SAME skinny kernels (bit-identical to the per-step
    //     launches) and the prep kernel gathers the row: three launches per forward gone.  LDX_EMB_TABLE=0 keeps the per-step launches.
    float* d_emb_table = nullptr;
    int build_emb_table();
    // (2) context cache (ldx_unet_context_cache): the caller promises that the bytes behind a ctx pointer do not change until it calls
    //     ldx_unet_context_cache again; the 16-bit copy of ctx and the batched k|v projection of every cross-attention (transformer.py:186-245 recomputes
    //     them every step only because a torch module has no notion of a sampling run) are then computed on the first evaluation of a (plan, ctx) only.
    bool ctx_cache = false; uint64_t ctx_epoch = 1;
    const void* kv_ptr = nullptr; uint64_t kv_epoch = 0;       // what the CURRENT plan's kvall buffer holds
    hipStream_t kv_stream = nullptr;                            // ... and the stream whose order it was filled in: a call on another stream refills (no cross-stream dependency is assumed)
    bool g_ctxc = false;                                        // the captured graph was recorded without the ctx_only ops
    int set_context_cache(int enable) { ctx_cache = enable != 0; ++ctx_epoch; return LDX_OK; }
    double algorithmic_flops() const;                           // steady_flops() + flops_shared
    double steady_flops() const;                                // algorithmic flops of one forward as executed in steady state (cached ops excluded)
    unsigned* d_sk_count = nullptr;                       // split-K tile counters (sk_counters()): one zeroed buffer per engine, every launch leaves it zeroed
    unsigned* sk_counters();
    // share > 0 (ldx_unet_denoise_cfg*): the evaluation batch is [uncond x share ; cond x share] over ONE latent batch x [share] — everything in front of the
    // first cross-attention (conv_in, the ResBlocks / Downsamples / self-attention up to it) sees identical inputs in both halves and is planned on `share`
    // samples; its results are copied into the second half's rows where the first cross-attention (and the skip connections) need them (OP_DUP).
    int plan(int B2, int h, int w, int Mc, int share = 0);
    int share_for(int B2, int h, int w, int xB, bool denoise, bool concat) const;
    int pShare = 0;
    int cfg_share = 1;                     // ldx_unet_cfg_share / LDX_CFG_SHARE: plan CFG evaluations with the shared prefix (0 never, 1 where it pays, 2 whenever possible)
    size_t prefix_end = 0;                 // plan(): number of ops in front of the first cross-attention (they ran on `share` samples)
    double flops_shared = 0;               // flops the current plan does NOT execute because of it (the second half's copy of the prefix ops)
    int64_t n_launches() const;
    int64_t n_graph_captures = 0, n_graph_replays = 0;      // ldx_graph_stats (tests: the sampler loops must replay, not re-capture)
    // per-kernel-class HIP-event profile of subsequent forwards (bench.py roofline leg)
    bool profiling = false, prof_detail = false;   // detail: key the report by op shape as well
    std::map<std::string, ProfEntry> prof;
    std::string profile_json() const;

    ldx_unet_config cfg;
    int device;
    DType dt;
    bool finalized = false;
    bool graph_mode = false;
    double flops = 0;
    size_t arena_cap = 0, arena_peak_dry = 0;
    size_t weight_bytes = 0;
    int pB2 = 0, ph = 0, pw = 0, pM = 0;

private:
    // weights
    std::unordered_map<std::string, HostTensor> host;
    std::string missing;
    std::vector<void*> dev_allocs;
    const HostTensor* get(const std::string& key, std::initializer_list<int64_t> shape);
    void* upload16(size_t rows, size_t cols, const std::function<float(size_t, size_t)>& getter);
    float* upload32(size_t n, const std::function<float(size_t)>& getter);
    bool mk_linear(const std::string& pre, int N, int K, bool bias, LinearW& out, bool conv1x1 = false);
    bool mk_conv3(const std::string& pre, int Cout, int Cin, int CinPad, LinearW& out);
    bool mk_norm(const std::string& pre, int C, NormW& out);
    bool mk_ln_folded(int N, int K, const std::function<float(size_t, size_t)>& W, const std::function<float(size_t)>& bias,
                      const std::string& norm_pre, LinearW& out, float*& c1);
    bool mk_res(const std::string& pre, int Cin, int Cout, ResW& r);
    bool mk_xf(const std::string& pre, int C, int depth, XfW& x);

    int exec_ops(hipStream_t ls, size_t op_begin = 0, size_t op_end = (size_t)-1, int ctx_sel = 0);      // ctx_sel: 0 all ops, 1 skip ctx_only ops, 2 ONLY ctx_only ops
    // GroupNorm workspace: gn_ws_rows producer rows per image + GN_FOLD folded rows, x 32 groups x 2 floats (ldx_kernels.h gn_workspace_rows)
    int gn_ws_rows = 256;
    size_t gn_ws_bytes(int B, long HWmax) { gn_ws_rows = (int)gn_workspace_rows(HWmax); return (size_t)B * (gn_ws_rows + GN_FOLD) * 32 * 2 * 4; }
    size_t ws_alloc(size_t bytes);                        // split-K workspace of one op (engine.cpp)
    void fuse_gn_stats();
    void fuse_gn_rowgemm();                               // GroupNorm (producer statistics) + proj_in -> one rowgemm launch
    bool op_rowgemm(const char* name, Act X, const LinearW& w, Act Y, Act R, int pro, const NormW* nw);          // post-pass over ops: GroupNorms whose input was just written by a fusable GEMM / conv get their statistics from its epilogue
    // per-call bindings read by exec_ops
    const float* b_x = nullptr; const float* b_s = nullptr; const float* b_ctx = nullptr; float* b_out = nullptr; bool b_den = false; int b_xB = 0, g_xB = 0;
    const float* b_cc = nullptr; const float* g_cc = nullptr; int b_ccn = 0, g_ccn = 0;
    const float* b_t = nullptr; const float* g_t = nullptr;       // caller-supplied timestep indices (run(): t_idx)
    const int* b_ids = nullptr; float* b_out2 = nullptr;
    // VAE
    std::vector<std::vector<ResW>> vae_up; std::vector<LinearW> vae_upconv; std::vector<bool> vae_has_up;
    ResW vae_mid1, vae_mid2; VaeAttnW vae_attn; NormW vae_norm_out; float* vae_pq = nullptr;
    // VAE encoder (optional: only when encoder.* weights were loaded)
    bool vae_has_enc = false; int vae_plan_mode = 0;      // 1 decode plan, 2 encode plan
    std::vector<std::vector<ResW>> enc_down; std::vector<LinearW> enc_downconv;
    ResW enc_mid1, enc_mid2; VaeAttnW enc_attn; NormW enc_norm_out; LinearW enc_conv_in, enc_conv_out; float* enc_qc = nullptr;
    bool mk_vae_attn(const std::string& pre, int C, VaeAttnW& a);
    // Flux
    std::vector<FluxDoubleW> fx_double; std::vector<FluxSingleW> fx_single;
    LinearW fx_img_in, fx_txt_in, fx_time0, fx_time1, fx_vec0, fx_vec1, fx_gd0, fx_gd1, fx_mod_all, fx_final;
    int fx_mod_total = 0, fx_final_mod_off = 0;
    std::vector<EmbSrc> fx_mod_srcs;
    const float *b_y = nullptr, *b_guid = nullptr, *b_cos = nullptr, *b_sin = nullptr;
    float *fx_temb = nullptr, *fx_gemb = nullptr, *fx_h1 = nullptr, *fx_vec = nullptr, *fx_svec = nullptr, *fx_mod = nullptr, *fx_tok = nullptr;
    // CLIP
    std::vector<ClipLayerW> clip_layers; NormW clip_final_ln; float* clip_tok = nullptr; float* clip_pos = nullptr;
    float* clip_extra = nullptr; int clip_extra_n = 0, clip_extra_cap = 0;     // textual-inversion rows for ids >= vocab_size
    float* clip_proj = nullptr;                                                // optional text_projection.weight [E][E] fp32 (CLIPTextModel.py:130,152-163)
    int clip_inter_planned = -100;
    LinearW te0, te2, conv_in, conv_out, emb_all;
    NormW out_gn;
    std::vector<BlockW> in_blocks, out_blocks;
    bool has_middle = false, mid_has_xf = false;
    ResW mid_res0, mid_res1;
    XfW mid_xf;
    int emb_total = 0;
    std::vector<EmbSrc> emb_srcs;
    // all cross-attention k|v projections of the context, batched into one GEMM per forward
    struct KvSrc { const HostTensor* k; const HostTensor* v; int C; };
    std::vector<KvSrc> kv_srcs; int kv_total = 0; LinearW kv_all; size_t kv_all_off = 0;
    float* d_log_sigmas = nullptr; float* d_temb = nullptr; int n_sigmas = 0;

    // plan
    void* arena = nullptr;
    std::vector<Op> ops;
    std::vector<std::pair<size_t, size_t>> free_list;
    std::map<size_t, size_t> live;
    size_t arena_top = 0, arena_peak = 0;
    size_t gn_ws_off = 0, prep_xc_off = 0;
    float *d_temb_out = nullptr, *d_e1 = nullptr, *d_e2 = nullptr, *d_emb_all = nullptr, *d_eps = nullptr;
    size_t a_alloc(size_t bytes);
    void a_free(size_t off);
    void* ptr(const Act& a) const { return (void*)((uintptr_t)arena + a.off + (size_t)a.col * 2); }
    Act new_act(int rows, int C);
    Act view(const Act& base, int col, int C);
    void release(const Act& a);
    void op_gemm(const char* name, Act A, const LinearW& w, Act C, Act R, bool geglu = false, const float* rowvec = nullptr, int rv_ld = 0, int rpb = 0);
    void op_conv(const char* name, Act X, int B, int Hin, int Win, int Cin, const LinearW& w, int stride, int Hout, int Wout,
                 Act Y, Act R, const float* rowvec = nullptr, int rv_ld = 0, float* Cf = nullptr, int ldcf = 0);
    void op_gn(const char* name, Act X, Act Y, int B, int HW, const NormW& n, float eps, bool silu);
    void op_ln(const char* name, Act X, Act Y, const NormW& n);
    void op_attn(const char* name, const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, Act O, int B, int H, int Nq, int Mk, int D);
    void emit_res(const ResW& r, Act X, Act OUT, int B, int H, int W);
    void emit_vae_attn(const VaeAttnW& a, Act X, Act OUT, int B, int H, int W);
    bool mk_vae_res(const std::string& pre, int Cin, int Cout, ResW& r);
    // Bshare > 0: the ops in front of the first cross-attention run on Bshare samples (see plan()); `dups` = views whose first Bshare * (their own H * W)
    // rows are to be copied into the following rows at that point (plus h itself)
    struct DupReq { Act a; int rows; size_t producer; };       // producer: index of the op that writes `a` (its dual store is preferred to a copy launch)
    void emit_xf(const XfW& x, Act X, Act OUT, int B, int H, int W, Act ctx16, int Mc, int Bshare = 0, const std::vector<DupReq>* dups = nullptr);
    void op_dup(const Act& a, int rows);          // rows [0, rows) of view a -> rows [rows, 2 rows)
    void dup_second_half(const DupReq& d);        // the producer's dual store (GemmArgs / RowGemmArgs::dup_rows) where it has one, else op_dup

    // UNet plans of other input shapes seen (multi-scale samplers alternate between two resolutions): launch plan, arena and
    // captured graph are kept per shape, so switching back costs nothing (a re-plan + two eager passes before the graph is
    // usable again cost ~17 ms per switch)
    struct PlanSnap {
        int B2 = 0, h = 0, w = 0, M = 0, share = 0; std::vector<Op> ops; double flops = 0, flops_shared = 0; void* arena = nullptr; size_t arena_cap = 0, arena_peak_dry = 0;
        size_t gn_ws_off = 0, prep_xc_off = 0, kv_all_off = 0; float *d_temb_out = nullptr, *d_e1 = nullptr, *d_e2 = nullptr, *d_emb_all = nullptr, *d_eps = nullptr;
        hipGraphExec_t graph_exec = nullptr; bool graph_valid = false, warm = false;
        const void* kv_ptr = nullptr; uint64_t kv_epoch = 0; hipStream_t kv_stream = nullptr; bool g_ctxc = false;
        const void *g_x = nullptr, *g_s = nullptr, *g_ctx = nullptr, *g_out = nullptr; bool g_den = false; int g_xB = 0; const float* g_cc = nullptr; int g_ccn = 0; const float* g_t = nullptr;
        // Flux plans: the per-shape buffers inside the arena and the first-block-cache op ranges
        float *fx_temb = nullptr, *fx_gemb = nullptr, *fx_h1 = nullptr, *fx_vec = nullptr, *fx_svec = nullptr, *fx_mod = nullptr, *fx_tok = nullptr;
        void *fb_s0 = nullptr, *fb_s1 = nullptr, *fb_x = nullptr; float *fb_first = nullptr, *fb_res = nullptr, *fb_part = nullptr;
        int fb_B = 0, fb_L = 0, fb_Lt = 0, fb_C = 0; size_t fb_a_end = 0, fb_b_end = 0;
    };
    std::vector<PlanSnap> plan_cache;
    void plan_stash();                    // move the current plan into plan_cache (evicting the oldest beyond 4)
    bool plan_restore(int B2, int h, int w, int Mc, int share = 0);
    std::vector<hipEvent_t> prof_events;
    bool prof_graph = false;
    // graph replay
    hipGraphExec_t graph_exec = nullptr;
    hipStream_t cap_stream = nullptr;
    bool graph_valid = false, warm = false;
    const void *g_x = nullptr, *g_s = nullptr, *g_ctx = nullptr, *g_out = nullptr; bool g_den = false;
};

}  // namespace ldx
