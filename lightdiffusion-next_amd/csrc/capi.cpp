// extern "C" surface of libldx.so (include/ldx.h).  No exceptions cross the ABI.
#include <cstring>
#include <mutex>
#include <new>

#include "engine.h"

using namespace ldx;

struct ldx_engine { Engine* impl; };

#define GUARD_BEGIN try {
#define GUARD_END                                                                  \
    }                                                                              \
    catch (const std::bad_alloc&) { set_error("out of host memory"); return LDX_EINVAL; } \
    catch (const std::exception& ex) { set_error(std::string("internal error: ") + ex.what()); return LDX_EINVAL; }

static int check_device(int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { set_error("no HIP device available (libldx has no CPU fallback)"); return LDX_EHIP; }
    if (device < 0 || device >= n) { set_error("device ordinal out of range"); return LDX_EINVAL; }
    return LDX_OK;
}
static inline DType dtype_of(int d) { return d == LDX_F16 ? DT_F16 : DT_BF16; }
static int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error(std::string(what) + ": " + hipGetErrorString(e)); return LDX_EHIP; }
    return LDX_OK;
}

// scratch for the single-op entry points (split-K partials); grown on demand, never shrunk
// One buffer PER DEVICE (the current one), growth under a mutex.  The single-op entry points are test / probe
// surface: calls that split K share this buffer, so they must be issued on ONE stream per device at a time
// (include/ldx.h says so); the engines own their workspaces and never come here.
static float* op_workspace(size_t floats) {
    static std::mutex mu;
    static float* buf[64] = {nullptr}; static size_t cap[64] = {0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    dev &= 63;
    std::lock_guard<std::mutex> lock(mu);
    if (floats > cap[dev]) {
        if (buf[dev]) { (void)hipDeviceSynchronize(); (void)hipFree(buf[dev]); buf[dev] = nullptr; cap[dev] = 0; }
        if (hipMalloc((void**)&buf[dev], floats * 4) != hipSuccess) return nullptr;
        cap[dev] = floats;
    }
    return buf[dev];
}
// tile counters of the in-kernel split-K reduction for the single-op entry points: per device, zeroed once, left at zero by every launch (same
// one-stream-per-device rule as the workspace)
static unsigned* op_sk_counters() {
    static std::mutex mu;
    static unsigned* buf[64] = {nullptr};
    int dev = 0;
    (void)hipGetDevice(&dev);
    dev &= 63;
    std::lock_guard<std::mutex> lock(mu);
    if (!buf[dev]) {
        if (hipMalloc((void**)&buf[dev], sizeof(unsigned) * SK_COUNTERS) != hipSuccess) { buf[dev] = nullptr; return nullptr; }
        if (hipMemset(buf[dev], 0, sizeof(unsigned) * SK_COUNTERS) != hipSuccess) { (void)hipFree(buf[dev]); buf[dev] = nullptr; return nullptr; }
    }
    return buf[dev];
}
static int attach_splitk(GemmArgs& g) {
    g.splitk = gemm_choose_splitk(g.M, g.N, g.K, g.geglu != 0);
    if (g.splitk > 1) {
        g.ws = op_workspace(gemm_sk_ws_floats(g.M, g.N, g.splitk));
        if (!g.ws) { set_error("split-K workspace allocation failed"); return LDX_EHIP; }
        g.sk_count = op_sk_counters();
    }
    return LDX_OK;
}

extern "C" {

const char* ldx_version(void) { return "ldx 0.1 (gfx950)"; }
const char* ldx_last_error(void) { return g_last_error.c_str(); }

int ldx_create(const ldx_unet_config* cfg, int device, ldx_engine** out) {
    GUARD_BEGIN
    if (!cfg || !out) { set_error("ldx_create: null argument"); return LDX_EINVAL; }
    int rc = check_device(device);
    if (rc) return rc;
    Engine* e = new Engine(*cfg, device);
    rc = e->validate();
    if (rc) { delete e; return rc; }
    *out = new ldx_engine{e};
    return LDX_OK;
    GUARD_END
}
void ldx_destroy(ldx_engine* e) { if (e) { delete e->impl; delete e; } }

int ldx_vae_create(const ldx_vae_config* cfg, int device, ldx_engine** out) {
    GUARD_BEGIN
    if (!cfg || !out) { set_error("ldx_vae_create: null argument"); return LDX_EINVAL; }
    int rc = check_device(device);
    if (rc) return rc;
    *out = new ldx_engine{new Engine(*cfg, device)};
    return LDX_OK;
    GUARD_END
}
int ldx_clip_create(const ldx_clip_config* cfg, int device, ldx_engine** out) {
    GUARD_BEGIN
    if (!cfg || !out) { set_error("ldx_clip_create: null argument"); return LDX_EINVAL; }
    int rc = check_device(device);
    if (rc) return rc;
    *out = new ldx_engine{new Engine(*cfg, device)};
    return LDX_OK;
    GUARD_END
}
int ldx_vae_decode(ldx_engine* e, const float* z, int B, int h, int w, float* out, void* stream) {
    GUARD_BEGIN
    if (!e) { set_error("null engine"); return LDX_EINVAL; }
    return e->impl->run_vae(z, B, h, w, out, (hipStream_t)stream);
    GUARD_END
}
int ldx_clip_encode(ldx_engine* e, const int32_t* ids, int B, int T, int inter_layer, float* out_last, float* out_inter, void* stream) {
    GUARD_BEGIN
    if (!e) { set_error("null engine"); return LDX_EINVAL; }
    return e->impl->run_clip((const int*)ids, B, T, inter_layer, out_last, out_inter, (hipStream_t)stream);
    GUARD_END
}
int ldx_clip_pooled(ldx_engine* e, const float* last, const int32_t* ids, int B, int T, int eos_token_id, float* out_pooled, void* stream) {
    GUARD_BEGIN
    if (!e) { set_error("null engine"); return LDX_EINVAL; }
    return e->impl->clip_pooled(last, (const int*)ids, B, T, eos_token_id, out_pooled, (hipStream_t)stream);
    GUARD_END
}
int ldx_clip_set_extra_embeddings(ldx_engine* e, const float* rows_host, int n) {
    GUARD_BEGIN
    if (!e) { set_error("null engine"); return LDX_EINVAL; }
    return e->impl->set_clip_extra(rows_host, n);
    GUARD_END
}
int ldx_esrgan_create(const ldx_esrgan_config* cfg, int device, ldx_engine** out) {
    GUARD_BEGIN
    if (!cfg || !out) { set_error("ldx_esrgan_create: null argument"); return LDX_EINVAL; }
    int rc = check_device(device);
    if (rc) return rc;
    *out = new ldx_engine{new Engine(*cfg, device)};
    return LDX_OK;
    GUARD_END
}
int ldx_esrgan_forward(ldx_engine* e, const float* pixels_nhwc, int B, int H, int W, float* out_nhwc, void* stream) {
    GUARD_BEGIN
    if (!e) { set_error("null engine"); return LDX_EINVAL; }
    return e->impl->run_esrgan(pixels_nhwc, B, H, W, out_nhwc, (hipStream_t)stream);
    GUARD_END
}
int ldx_tile_blend(const float* tile, int th, int tw, float* out, float* div, int H, int W, int C, int y0, int x0, int feather, void* stream) {
    if (!tile || !out || !div || th <= 0 || tw <= 0 || C <= 0 || y0 < 0 || x0 < 0 || y0 + th > H || x0 + tw > W || feather < 0) {
        set_error("ldx_tile_blend: bad argument (tile must lie inside the output)"); return LDX_EINVAL; }
    launch_tile_blend(tile, th, tw, out, div, H, W, C, y0, x0, feather, (hipStream_t)stream);
    return check_launch("ldx_tile_blend");
}
int ldx_tile_finish(float* out, const float* div, int64_t n, int clamp01, void* stream) {
    if (!out || n < 0) { set_error("ldx_tile_finish: bad argument"); return LDX_EINVAL; }
    launch_tile_finish(out, div, (size_t)n, clamp01, (hipStream_t)stream);
    return check_launch("ldx_tile_finish");
}
int ldx_t5_create(const ldx_t5_config* cfg, int device, ldx_engine** out) {
    GUARD_BEGIN
    if (!cfg || !out) { set_error("ldx_t5_create: null argument"); return LDX_EINVAL; }
    int rc = check_device(device);
    if (rc) return rc;
    *out = new ldx_engine{new Engine(*cfg, device)};
    return LDX_OK;
    GUARD_END
}
int ldx_t5_encode(ldx_engine* e, const int32_t* ids, int B, int L, const float* bias, float* out, void* stream) {
    GUARD_BEGIN
    if (!e) { set_error("null engine"); return LDX_EINVAL; }
    return e->impl->run_t5((const int*)ids, B, L, bias, out, (hipStream_t)stream);
    GUARD_END
}
int ldx_flux_fbcache(ldx_engine* e, float residual_diff_threshold) {
    GUARD_BEGIN
    if (!e || e->impl->kind != KIND_FLUX) { set_error("ldx_flux_fbcache: not a Flux engine"); return LDX_EINVAL; }
    e->impl->fb_threshold = residual_diff_threshold > 0.f ? residual_diff_threshold : 0.f;
    e->impl->fb_reset(); e->impl->fb_hits = e->impl->fb_misses = 0;
    return LDX_OK;
    GUARD_END
}
int ldx_flux_set_fp8(ldx_engine* e, int enable) {
    GUARD_BEGIN
    if (!e || e->impl->kind != KIND_FLUX) { set_error("ldx_flux_set_fp8: not a Flux engine"); return LDX_EINVAL; }
    if (e->impl->finalized) { set_error("ldx_flux_set_fp8: call before ldx_finalize (the weights are quantised there)"); return LDX_ESTATE; }
    if (enable < 0 || enable > 3) { set_error("ldx_flux_set_fp8: mode must be 0 (off), 1 / 2 (linears) or 3 (linears + attention)"); return LDX_EINVAL; }
    e->impl->fx_fp8 = enable != 0;
    e->impl->fx_fp8_attn = enable == 3;        // 1 (= 2): the linears only, QK^T / PV stay 16-bit — what the mode has meant since round 2; 3: linears AND attention (explicit opt-in)
    return LDX_OK;
    GUARD_END
}
int ldx_flux_fbcache_stats(ldx_engine* e, int64_t* hits, int64_t* misses) {
    GUARD_BEGIN
    if (!e || e->impl->kind != KIND_FLUX) { set_error("ldx_flux_fbcache_stats: not a Flux engine"); return LDX_EINVAL; }
    if (hits) *hits = e->impl->fb_hits;
    if (misses) *misses = e->impl->fb_misses;
    return LDX_OK;
    GUARD_END
}
int ldx_flux_create(const ldx_flux_config* cfg, int device, ldx_engine** out) {
    GUARD_BEGIN
    if (!cfg || !out) { set_error("ldx_flux_create: null argument"); return LDX_EINVAL; }
    int rc = check_device(device);
    if (rc) return rc;
    *out = new ldx_engine{new Engine(*cfg, device)};
    return LDX_OK;
    GUARD_END
}
int ldx_flux_forward(ldx_engine* e, const float* x, const float* sigma, const float* ctx, const float* y, const float* guidance,
                     const float* pe_cos, const float* pe_sin, int B, int h, int w, int Lt, int denoise, float* out, void* stream) {
    GUARD_BEGIN
    if (!e) { set_error("null engine"); return LDX_EINVAL; }
    return e->impl->run_flux(x, sigma, ctx, y, guidance, pe_cos, pe_sin, B, h, w, Lt, denoise != 0, out, (hipStream_t)stream);
    GUARD_END
}
int ldx_load_tensor(ldx_engine* e, const char* key, const void* data, int dtype, const int64_t* shape, int ndim) {
    GUARD_BEGIN
    if (!e) { set_error("null engine"); return LDX_EINVAL; }
    return e->impl->load_tensor(key, data, dtype, shape, ndim);
    GUARD_END
}
int ldx_set_tables(ldx_engine* e, const float* log_sigmas, int n, const float* temb, int temb_dim) {
    GUARD_BEGIN
    if (!e) { set_error("null engine"); return LDX_EINVAL; }
    return e->impl->set_tables(log_sigmas, n, temb, temb_dim);
    GUARD_END
}
int ldx_finalize(ldx_engine* e) {
    GUARD_BEGIN
    if (!e) { set_error("null engine"); return LDX_EINVAL; }
    if (e->impl->kind == KIND_VAE) return e->impl->finalize_vae();
    if (e->impl->kind == KIND_CLIP) return e->impl->finalize_clip();
    if (e->impl->kind == KIND_FLUX) return e->impl->finalize_flux();
    if (e->impl->kind == KIND_T5) return e->impl->finalize_t5();
    if (e->impl->kind == KIND_ESRGAN) return e->impl->finalize_esrgan();
    return e->impl->finalize();
    GUARD_END
}
int ldx_unet_denoise(ldx_engine* e, const float* x, const float* sigma, const float* ctx, int B2, int h, int w, int M, float* out, void* stream) {
    GUARD_BEGIN
    if (!e) { set_error("null engine"); return LDX_EINVAL; }
    if (e->impl->kind != KIND_UNET) { set_error("ldx_unet_denoise: not a UNet engine"); return LDX_ESTATE; }
    return e->impl->run(x, sigma, ctx, B2, h, w, M, out, true, (hipStream_t)stream);
    GUARD_END
}
int ldx_unet_denoise_concat(ldx_engine* e, const float* x, const float* sigma, const float* ctx, const float* c_concat, int cc_channels, int B2, int h, int w, int M,
                            float* out, void* stream) {
    GUARD_BEGIN
    if (!e) { set_error("null engine"); return LDX_EINVAL; }
    if (e->impl->kind != KIND_UNET) { set_error("ldx_unet_denoise_concat: not a UNet engine"); return LDX_ESTATE; }
    if (!c_concat) { set_error("ldx_unet_denoise_concat: c_concat is null (use ldx_unet_denoise)"); return LDX_EINVAL; }
    return e->impl->run(x, sigma, ctx, B2, h, w, M, out, true, (hipStream_t)stream, 0, c_concat, cc_channels);
    GUARD_END
}
int ldx_unet_denoise_cfg(ldx_engine* e, const float* x, float sigma, const float* ctx, int B, int h, int w, int M, float* out, void* stream) {
    GUARD_BEGIN
    if (!e) { set_error("null engine"); return LDX_EINVAL; }
    if (e->impl->kind != KIND_UNET) { set_error("ldx_unet_denoise_cfg: not a UNet engine"); return LDX_ESTATE; }
    return e->impl->run_cfg(x, sigma, ctx, B, h, w, M, out, (hipStream_t)stream);
    GUARD_END
}
int ldx_unet_denoise_cfg_t(ldx_engine* e, const float* x, float sigma, int t_index, const float* ctx, int B, int h, int w, int M, float* out, void* stream) {
    GUARD_BEGIN
    if (!e) { set_error("null engine"); return LDX_EINVAL; }
    if (e->impl->kind != KIND_UNET) { set_error("ldx_unet_denoise_cfg_t: not a UNet engine"); return LDX_ESTATE; }
    return e->impl->run_cfg(x, sigma, ctx, B, h, w, M, out, (hipStream_t)stream, t_index < 0 ? -1 : t_index);
    GUARD_END
}
int ldx_unet_denoise_t(ldx_engine* e, const float* x, const float* sigma, const float* t_index, const float* ctx, int B2, int h, int w, int M, float* out, void* stream) {
    GUARD_BEGIN
    if (!e) { set_error("null engine"); return LDX_EINVAL; }
    if (e->impl->kind != KIND_UNET) { set_error("ldx_unet_denoise_t: not a UNet engine"); return LDX_ESTATE; }
    if (!t_index) { set_error("ldx_unet_denoise_t: t_index is null (use ldx_unet_denoise)"); return LDX_EINVAL; }
    return e->impl->run(x, sigma, ctx, B2, h, w, M, out, true, (hipStream_t)stream, 0, nullptr, 0, t_index);
    GUARD_END
}
int ldx_unet_timestep(ldx_engine* e, const float* sigma, int n, int32_t* t_out, void* stream) {
    GUARD_BEGIN
    if (!e) { set_error("null engine"); return LDX_EINVAL; }
    return e->impl->timestep_lookup(sigma, n, (int*)t_out, (hipStream_t)stream);
    GUARD_END
}
int ldx_unet_forward(ldx_engine* e, const float* x, const float* t, const float* ctx, int B2, int h, int w, int M, float* out, void* stream) {
    GUARD_BEGIN
    if (!e) { set_error("null engine"); return LDX_EINVAL; }
    if (e->impl->kind != KIND_UNET) { set_error("ldx_unet_forward: not a UNet engine"); return LDX_ESTATE; }
    return e->impl->run(x, t, ctx, B2, h, w, M, out, false, (hipStream_t)stream);
    GUARD_END
}
int ldx_plan_info(ldx_engine* e, int64_t* n_launches, double* flops, int64_t* arena_bytes) {
    if (!e) { set_error("null engine"); return LDX_EINVAL; }
    if (n_launches) *n_launches = e->impl->n_launches();
    if (flops) *flops = e->impl->algorithmic_flops();
    if (arena_bytes) *arena_bytes = (int64_t)e->impl->arena_cap;
    return LDX_OK;
}
int ldx_plan_flops(ldx_engine* e, double* executed, double* shared) {
    if (!e) { set_error("null engine"); return LDX_EINVAL; }
    if (executed) *executed = e->impl->steady_flops();
    if (shared) *shared = e->impl->flops_shared;
    return LDX_OK;
}
int ldx_unet_cfg_share(ldx_engine* e, int enable) {
    if (!e) { set_error("null engine"); return LDX_EINVAL; }
    if (e->impl->kind != KIND_UNET) { set_error("ldx_unet_cfg_share: not a UNet engine"); return LDX_ESTATE; }
    e->impl->cfg_share = enable < 0 ? 0 : (enable > 2 ? 2 : enable);
    return LDX_OK;
}
int ldx_profile(ldx_engine* e, int enable, int reset) {
    if (!e) { set_error("null engine"); return LDX_EINVAL; }
    e->impl->profiling = enable != 0;
    e->impl->prof_detail = enable == 2;
    if (reset) e->impl->prof.clear();
    return LDX_OK;
}
int ldx_profile_report(ldx_engine* e, char* buf, int64_t cap) {
    if (!e || !buf || cap <= 0) { set_error("ldx_profile_report: bad argument"); return LDX_EINVAL; }
    const std::string s = e->impl->profile_json();
    if ((int64_t)s.size() + 1 > cap) { set_error("ldx_profile_report: buffer too small"); return LDX_EINVAL; }
    memcpy(buf, s.c_str(), s.size() + 1);
    return LDX_OK;
}
int ldx_set_graph_mode(ldx_engine* e, int enable) {
    if (!e) { set_error("null engine"); return LDX_EINVAL; }
    e->impl->graph_mode = enable != 0;
    return LDX_OK;
}
int ldx_unet_context_cache(ldx_engine* e, int enable) {
    if (!e) { set_error("null engine"); return LDX_EINVAL; }
    if (e->impl->kind != KIND_UNET) { set_error("ldx_unet_context_cache: not a UNet engine"); return LDX_ESTATE; }
    return e->impl->set_context_cache(enable);
}
int ldx_reload_env(void) {
    reload_dispatch_env();
    return LDX_OK;
}
int ldx_graph_stats(ldx_engine* e, int64_t* captures, int64_t* replays) {
    if (!e || !captures || !replays) { set_error("ldx_graph_stats: bad argument"); return LDX_EINVAL; }
    *captures = e->impl->n_graph_captures; *replays = e->impl->n_graph_replays;
    return LDX_OK;
}

int ldx_sampler_step(int kind, float* x, const float* du, const float* dc, float* dout, int64_t n, float cfg, float c0, float c1, void* stream) {
    if (!x || !du || !dc || n < 0 || (kind < 0 || kind > 3)) { set_error("ldx_sampler_step: bad argument"); return LDX_EINVAL; }
    if (kind == 2 && !dout) { set_error("ldx_sampler_step: kind 2 (CFG combine only) needs denoised_out"); return LDX_EINVAL; }
    if (kind == 0 && c0 == 0.f) { set_error("ldx_sampler_step: kind 0 (Euler) divides by c0 = sigma_hat, which must be non-zero"); return LDX_EINVAL; }
    StepArgs a{x, du, dc, dout, (size_t)n, cfg, kind, c0, c1};
    launch_sampler_step(a, (hipStream_t)stream);
    return check_launch("ldx_sampler_step");
}
int ldx_bislerp_pass(const float* in, float* out, int N, int C, int H, int W, int axis, int new_len,
                     const int32_t* c1, const int32_t* c2, const float* ratios, void* stream) {
    if (!in || !out || !c1 || !c2 || !ratios || N <= 0 || C <= 0 || C > 16 || H <= 0 || W <= 0 || new_len <= 0 || (axis != 0 && axis != 1)) {
        set_error("ldx_bislerp_pass: bad argument (C <= 16, axis 0|1)"); return LDX_EINVAL; }
    launch_bislerp_pass(in, out, N, C, H, W, axis, new_len, (const int*)c1, (const int*)c2, ratios, (hipStream_t)stream);
    return check_launch("ldx_bislerp_pass");
}
int ldx_vae_encode(ldx_engine* e, const float* pixels_nhwc, int B, int H, int W, float* moments_nchw, void* stream) {
    GUARD_BEGIN
    if (!e) { set_error("null engine"); return LDX_EINVAL; }
    return e->impl->run_vae_encode(pixels_nhwc, B, H, W, moments_nchw, (hipStream_t)stream);
    GUARD_END
}
int ldx_bilinear(const float* in, float* out, int planes, int hin, int win, int hout, int wout, void* stream) {
    if (!in || !out || planes <= 0 || hin <= 0 || win <= 0 || hout <= 0 || wout <= 0) { set_error("ldx_bilinear: bad argument"); return LDX_EINVAL; }
    launch_bilinear(in, out, planes, hin, win, hout, wout, (hipStream_t)stream);
    return check_launch("ldx_bilinear");
}

int ldx_op_convert(const float* in_f32, void* out_16, int64_t n, int dtype, int to_f32, void* stream) {
    if (!in_f32 || !out_16 || n < 0) { set_error("ldx_op_convert: bad argument"); return LDX_EINVAL; }
    if (to_f32) launch_t_to_f32(out_16, (float*)in_f32, (size_t)n, dtype_of(dtype), (hipStream_t)stream);
    else launch_f32_to_t(in_f32, out_16, (size_t)n, dtype_of(dtype), (hipStream_t)stream);
    return check_launch("ldx_op_convert");
}
int ldx_op_gemm(const void* A, int lda, const void* W, int M, int N, int K, const float* bias, const float* rowvec, int rowvec_ld,
                int rows_per_batch, int geglu, const void* R, int ldr, void* C, int ldc, float* Cf, int ldcf, int dtype, void* stream) {
    if (!A || !W || M <= 0 || N <= 0 || K <= 0 || K % 8 || lda % 8 || (!C && !Cf)) { set_error("ldx_op_gemm: bad argument (K % 8, lda % 8)"); return LDX_EINVAL; }
    GemmArgs g{};
    g.A = A; g.lda = lda; g.W = W; g.M = M; g.N = N; g.K = K; g.mode = 0; g.bias = bias;
    g.rowvec = rowvec; g.rowvec_ld = rowvec_ld; g.rows_per_batch = rows_per_batch > 0 ? rows_per_batch : 1; g.geglu = geglu;
    g.R = R; g.ldr = ldr; g.C = C; g.ldc = ldc; g.Cf = Cf; g.ldcf = ldcf;
    if (int rc = attach_splitk(g)) return rc;
    launch_gemm(g, dtype_of(dtype), (hipStream_t)stream);
    return check_launch("ldx_op_gemm");
}
int ldx_op_layernorm_mx(const void* X, int ldx_, int rows, int C, float eps, const float* gamma, const float* beta,
                        void* Y8, int ldy8, void* S8, int s8_ld, int dtype, void* stream) {
    if (!X || !Y8 || !S8 || !gamma || !beta || rows <= 0 || C % 128 || C > 4096 || ldx_ % 8 || ldy8 % 16 || ldy8 < C || s8_ld < rows) {
        set_error("ldx_op_layernorm_mx: bad argument (C % 128, C <= 4096, ldy8 % 16, s8_ld >= rows)"); return LDX_EINVAL; }
    LayerNormArgs a{X, ldx_, nullptr, 0, rows, C, eps, gamma, beta};
    a.Y8 = Y8; a.ldy8 = ldy8; a.S8 = (uint32_t*)S8; a.s8_ld = s8_ld;
    launch_layernorm(a, dtype_of(dtype), (hipStream_t)stream);
    return check_launch("ldx_op_layernorm_mx");
}
int ldx_op_attention_mx(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, void* O8, int ldo8, void* SO, int so_ld,
                        int B, int H, int Nq, int Mk, float scale, int dtype, void* stream) {
    AttnArgs a{};
    a.Q = Q; a.ldq = ldq; a.K = K; a.ldk = ldk; a.V = V; a.ldv = ldv; a.B = B; a.H = H; a.Nq = Nq; a.Mk = Mk; a.D = 128; a.scale = scale;
    a.O8 = O8; a.ldo8 = ldo8; a.SO = (uint32_t*)SO; a.so_ld = so_ld;
    if (!Q || !K || !V || !O8 || !SO || B <= 0 || H <= 0 || Nq <= 0 || Mk <= 0 || ldq % 8 || ldk % 8 || ldv % 8 || ldo8 % 16 || ldo8 < H * 128 || so_ld < B * Nq) {
        set_error("ldx_op_attention_mx: bad argument"); return LDX_EINVAL; }
    if (attn_pipe128_ok(a)) a.knorm_ws = op_workspace((size_t)B * H * ((Mk + 63) / 64));
    if (!attention_mx_out_ok(a)) { set_error("ldx_op_attention_mx: needs head dim 128 and at least 16 query blocks of 128 (B * H * ceil(Nq / 128))"); return LDX_EINVAL; }
    launch_attention(a, dtype_of(dtype), (hipStream_t)stream);
    return check_launch("ldx_op_attention_mx");
}
int ldx_op_mx_vt_quant(const void* V, int ldv, int B, int H, int L, void* V8T, void* SV, int Lp, int dtype, void* stream) {
    if (!V || !V8T || !SV || B <= 0 || H <= 0 || L <= 0 || ldv % 8 || Lp % 128 || Lp < L) { set_error("ldx_op_mx_vt_quant: bad argument (ldv % 8, Lp % 128 == 0, Lp >= L)"); return LDX_EINVAL; }
    MxVtArgs a{V, ldv, B, H, L, V8T, (uint32_t*)SV, Lp};
    launch_mx_vt_quant(a, dtype_of(dtype), (hipStream_t)stream);
    return check_launch("ldx_op_mx_vt_quant");
}
int ldx_op_attention_fp8(const void* Q8, int ldq8, const void* SQ, int sq_ld, const void* K8, int ldk8, const void* SK, int sk_ld, const void* V8T, const void* SV, int Lp,
                         void* O, int ldo, void* O8, int ldo8, void* SO, int so_ld, int B, int H, int Nq, int Mk, float scale, int dtype, void* stream) {
    AttnMxArgs a{};
    a.Q8 = Q8; a.ldq8 = ldq8; a.SQ = (const uint32_t*)SQ; a.sq_ld = sq_ld; a.K8 = K8; a.ldk8 = ldk8; a.SK = (const uint32_t*)SK; a.sk_ld = sk_ld;
    a.V8T = V8T; a.SV = (const uint32_t*)SV; a.Lp = Lp; a.O = O; a.ldo = ldo; a.O8 = O8; a.ldo8 = ldo8; a.SO = (uint32_t*)SO; a.so_ld = so_ld;
    a.B = B; a.H = H; a.Nq = Nq; a.Mk = Mk; a.scale = scale;
    if (!attn_mx_ok(a) || sq_ld < B * Nq || sk_ld < B * Mk) { set_error("ldx_op_attention_fp8: bad argument (head dim 128; ld % 16; Lp % 128 == 0, Lp >= Mk; scale strides >= rows)"); return LDX_EINVAL; }
    launch_attn_mx(a, dtype_of(dtype), (hipStream_t)stream);
    return check_launch("ldx_op_attention_fp8");
}
int ldx_op_qk_norm_rope_mx(const void* QKV, int ld, int rows, int L, int H, const float* qscale, const float* kscale, const float* cosT, const float* sinT, float eps,
                           void* Q8, void* K8, int ld8, void* SQ, void* SK, int s_ld, int dtype, void* stream) {
    if (!QKV || !qscale || !kscale || !cosT || !sinT || !Q8 || !K8 || !SQ || !SK || rows <= 0 || L <= 0 || H <= 0 || ld % 8 || ld8 % 16 || ld8 < H * 128 || s_ld < rows) {
        set_error("ldx_op_qk_norm_rope_mx: bad argument (head dim 128; ld % 8, ld8 % 16, s_ld >= rows)"); return LDX_EINVAL; }
    QkRopeArgs a{};
    a.QKV = (void*)QKV; a.ld = ld; a.rows = rows; a.L = L; a.H = H; a.D = 128; a.qscale = qscale; a.kscale = kscale; a.cosT = cosT; a.sinT = sinT; a.eps = eps;
    a.Q8 = Q8; a.K8 = K8; a.ld8 = ld8; a.SQ = (uint32_t*)SQ; a.SK = (uint32_t*)SK; a.s8_ld = s_ld; a.row8 = 0;
    launch_qk_norm_rope_mx(a, dtype_of(dtype), (hipStream_t)stream);
    return check_launch("ldx_op_qk_norm_rope_mx");
}
int ldx_op_mx_quant(const void* X, int ldx_, int rows, int K, void* Y, int ldy, void* scales, int scales_ld, int dtype, void* stream) {
    if (!X || !Y || !scales || rows <= 0 || K <= 0 || K % 128 || ldx_ % 8 || ldy % 16 || ldy < K || scales_ld < rows) {
        set_error("ldx_op_mx_quant: bad argument (K % 128, ldx % 8, ldy % 16, scales_ld >= rows)"); return LDX_EINVAL; }
    MxQuantArgs a{X, ldx_, rows, K, Y, ldy, (uint32_t*)scales, scales_ld};
    launch_mx_quant(a, dtype_of(dtype), (hipStream_t)stream);
    return check_launch("ldx_op_mx_quant");
}
int ldx_op_gemm_mx(const void* A8, int lda, const void* SA, int sa_ld, const void* W8, const void* SW, int sw_ld, int M, int N, int K,
                   const float* bias, int act, const void* R, int ldr, void* C, int ldc, float* Cf, int ldcf,
                   void* C8, int ldc8, void* SC, int sc_ld, int dtype, void* stream) {
    if (!A8 || !W8 || !SA || !SW || M <= 0 || N <= 0 || K <= 0 || K % 128 || lda % 16 || sa_ld < M || sw_ld < N || (!C && !Cf && !C8) || act < 0 || act > 3) {
        set_error("ldx_op_gemm_mx: bad argument (K % 128, lda % 16, sa_ld >= M, sw_ld >= N)"); return LDX_EINVAL; }
    if (C8 && (!SC || N % 128 || ldc8 % 16 || ldc8 < N || sc_ld < M || R || C || Cf)) {
        set_error("ldx_op_gemm_mx: MX output needs N % 128 == 0, ldc8 % 16 == 0, scales_ld >= M and no other output / residual"); return LDX_EINVAL; }
    GemmArgs g{};
    g.A = A8; g.lda = lda; g.W = W8; g.M = M; g.N = N; g.K = K; g.mode = 0; g.bias = bias; g.rows_per_batch = 1; g.act = act;
    g.f8 = 1; g.SA = (const uint32_t*)SA; g.sa_ld = sa_ld; g.SW = (const uint32_t*)SW; g.sw_ld = sw_ld;
    g.R = R; g.ldr = ldr; g.C = C; g.ldc = ldc; g.Cf = Cf; g.ldcf = ldcf;
    g.C8 = C8; g.ldc8 = ldc8; g.c8_col = 0; g.SC = (uint32_t*)SC; g.sc_ld = sc_ld;
    g.splitk = C8 ? 1 : gemm_choose_splitk(M, N, K / 2, false);
    if (g.splitk > 1) {
        g.ws = op_workspace(gemm_sk_ws_floats(M, N, g.splitk));
        if (!g.ws) { set_error("split-K workspace allocation failed"); return LDX_EHIP; }
        g.sk_count = op_sk_counters();
    }
    launch_gemm(g, dtype_of(dtype), (hipStream_t)stream);
    return check_launch("ldx_op_gemm_mx");
}
int ldx_op_gemm2(const void* A1, int lda1, const void* W1, int M1, int N1, int K1, const float* bias1, void* C1, int ldc1,
                 const void* A2, int lda2, const void* W2, int M2, int N2, int K2, const float* bias2, void* C2, int ldc2, int dtype, void* stream) {
    if (!A1 || !W1 || !C1 || !A2 || !W2 || !C2 || M1 <= 0 || N1 <= 0 || K1 <= 0 || M2 <= 0 || N2 <= 0 || K2 <= 0 || K1 % 8 || K2 % 8 || lda1 % 8 || lda2 % 8) {
        set_error("ldx_op_gemm2: bad argument (K % 8, lda % 8)"); return LDX_EINVAL; }
    GemmArgs a{}, b{};
    a.A = A1; a.lda = lda1; a.W = W1; a.M = M1; a.N = N1; a.K = K1; a.bias = bias1; a.C = C1; a.ldc = ldc1; a.rows_per_batch = 1;
    b.A = A2; b.lda = lda2; b.W = W2; b.M = M2; b.N = N2; b.K = K2; b.bias = bias2; b.C = C2; b.ldc = ldc2; b.rows_per_batch = 1;
    launch_gemm2(a, b, dtype_of(dtype), (hipStream_t)stream);
    return check_launch("ldx_op_gemm2");
}
int ldx_op_gemm2_mx(const void* A1, int lda1, const void* SA1, int sa_ld1, const void* W1, const void* SW1, int sw_ld1, int M1, int N1, int K1,
                    const float* bias1, void* C1, int ldc1,
                    const void* A2, int lda2, const void* SA2, int sa_ld2, const void* W2, const void* SW2, int sw_ld2, int M2, int N2, int K2,
                    const float* bias2, void* C2, int ldc2, int dtype, void* stream) {
    if (!A1 || !W1 || !SA1 || !SW1 || !C1 || !A2 || !W2 || !SA2 || !SW2 || !C2 || M1 <= 0 || N1 <= 0 || K1 <= 0 || M2 <= 0 || N2 <= 0 || K2 <= 0 ||
        K1 % 128 || K2 % 128 || lda1 % 16 || lda2 % 16 || sa_ld1 < M1 || sw_ld1 < N1 || sa_ld2 < M2 || sw_ld2 < N2) {
        set_error("ldx_op_gemm2_mx: bad argument (K % 128, lda % 16, sa_ld >= M, sw_ld >= N)"); return LDX_EINVAL; }
    GemmArgs a{}, b{};
    a.A = A1; a.lda = lda1; a.W = W1; a.M = M1; a.N = N1; a.K = K1; a.bias = bias1; a.C = C1; a.ldc = ldc1; a.rows_per_batch = 1;
    a.f8 = 1; a.SA = (const uint32_t*)SA1; a.sa_ld = sa_ld1; a.SW = (const uint32_t*)SW1; a.sw_ld = sw_ld1;
    b.A = A2; b.lda = lda2; b.W = W2; b.M = M2; b.N = N2; b.K = K2; b.bias = bias2; b.C = C2; b.ldc = ldc2; b.rows_per_batch = 1;
    b.f8 = 1; b.SA = (const uint32_t*)SA2; b.sa_ld = sa_ld2; b.SW = (const uint32_t*)SW2; b.sw_ld = sw_ld2;
    launch_gemm2(a, b, dtype_of(dtype), (hipStream_t)stream);
    return check_launch("ldx_op_gemm2_mx");
}
int ldx_op_conv3x3(const void* X, int ldx_, const void* W, int B, int Hin, int Win, int Cin, int Cout, int stride, int Hout, int Wout,
                   int resize_to_out, const float* bias, const float* rowvec, int rowvec_ld, const void* R, int ldr, void* Y, int ldy,
                   int dtype, void* stream) {
    if (!X || !W || !Y || Cin % 64 || ldx_ % 8 || (stride != 1 && stride != 2)) { set_error("ldx_op_conv3x3: bad argument (Cin % 64)"); return LDX_EINVAL; }
    GemmArgs g{};
    g.A = X; g.lda = ldx_; g.W = W; g.M = B * Hout * Wout; g.N = Cout; g.K = 9 * Cin; g.mode = 1;
    g.Cin = Cin; g.Hin = Hin; g.Win = Win; g.Hout = Hout; g.Wout = Wout; g.stride = stride;
    if (resize_to_out) { g.Hv = Hout; g.Wv = Wout; } else { g.Hv = Hin; g.Wv = Win; }
    g.resize = (g.Hv != Hin || g.Wv != Win) ? 1 : 0;
    g.bias = bias; g.rowvec = rowvec; g.rowvec_ld = rowvec_ld; g.rows_per_batch = Hout * Wout;
    g.R = R; g.ldr = ldr; g.C = Y; g.ldc = ldy;
    if (int rc = attach_splitk(g)) return rc;
    launch_gemm(g, dtype_of(dtype), (hipStream_t)stream);
    return check_launch("ldx_op_conv3x3");
}
int ldx_op_conv3x3_skip(const void* X, int ldx_, const void* X2, int ldx2, int Cin2, const void* W, int B, int H, int Wd, int Cin, int Cout,
                        const float* bias, void* Y, int ldy, int dtype, void* stream) {
    if (!X || !X2 || !W || !Y || Cin % 64 || Cin2 % 64 || ldx_ % 8 || ldx2 % 8 || B <= 0 || H <= 0 || Wd <= 0) {
        set_error("ldx_op_conv3x3_skip: bad argument (Cin % 64, Cin2 % 64)"); return LDX_EINVAL; }
    GemmArgs g{};
    g.A = X; g.lda = ldx_; g.W = W; g.M = B * H * Wd; g.N = Cout; g.K = 9 * Cin + Cin2; g.mode = 1;
    g.Cin = Cin; g.Hin = H; g.Win = Wd; g.Hout = H; g.Wout = Wd; g.stride = 1; g.Hv = H; g.Wv = Wd; g.resize = 0;
    g.A2 = X2; g.lda2 = ldx2; g.Cin2 = Cin2;
    g.bias = bias; g.rows_per_batch = H * Wd; g.C = Y; g.ldc = ldy;
    if (int rc = attach_splitk(g)) return rc;
    launch_gemm(g, dtype_of(dtype), (hipStream_t)stream);
    return check_launch("ldx_op_conv3x3_skip");
}
int64_t ldx_op_groupnorm_workspace_floats(int B, int G) { return (int64_t)B * GN_NCHUNK * G * 2; }
int ldx_op_groupnorm(const void* X, int ldx_, void* Y, int ldy, int B, int HW, int C, int G, float eps, int silu,
                     const float* gamma, const float* beta, float* workspace, int dtype, void* stream) {
    if (!X || !Y || !gamma || !beta || !workspace || C % 8 || C % G || G > 32 || ldx_ % 8 || ldy % 8) { set_error("ldx_op_groupnorm: bad argument"); return LDX_EINVAL; }
    GroupNormArgs a{X, ldx_, Y, ldy, B, HW, C, G, eps, silu, gamma, beta, workspace};
    launch_groupnorm(a, dtype_of(dtype), (hipStream_t)stream);
    return check_launch("ldx_op_groupnorm");
}
int ldx_op_layernorm(const void* X, int ldx_, void* Y, int ldy, int rows, int C, float eps, const float* gamma, const float* beta, int dtype, void* stream) {
    if (!X || !Y || !gamma || !beta || C % 8 || C > 4096) { set_error("ldx_op_layernorm: bad argument (C % 8, C <= 4096)"); return LDX_EINVAL; }
    LayerNormArgs a{X, ldx_, Y, ldy, rows, C, eps, gamma, beta};
    launch_layernorm(a, dtype_of(dtype), (hipStream_t)stream);
    return check_launch("ldx_op_layernorm");
}
int ldx_op_attention(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, void* O, int ldo, int B, int H, int Nq, int Mk, int D,
                     float scale, int causal, int dtype, void* stream) {
    if (!Q || !K || !V || !O || D % 8 || (D > 160 && D != 512) || D <= 0 || ldq % 8 || ldk % 8 || ldv % 8 || ldo % 4 || Mk <= 0) { set_error("ldx_op_attention: bad argument (D % 8, D <= 160 or D == 512)"); return LDX_EINVAL; }
    AttnArgs a{Q, ldq, K, ldk, V, ldv, O, ldo, B, H, Nq, Mk, D, scale, causal, nullptr, 0, 0};
    if (D == 512) {
        if (!attn512_ok(a)) { set_error("ldx_op_attention: D = 512 without a mask only (attn512.hip)"); return LDX_EINVAL; }
        a.nsplit = attn512_splits(a);
        if (a.nsplit > 1) { a.split_ws = op_workspace(attn512_ws_floats(a, a.nsplit)); if (!a.split_ws) { set_error("attention split workspace allocation failed"); return LDX_EHIP; } }
    }
    if (attn_pipe_ok(a) || attn_pipe128_ok(a)) a.knorm_ws = op_workspace((size_t)B * H * ((Mk + 63) / 64));      // (shared single-op scratch: one stream per device, include/ldx.h)
    launch_attention(a, dtype_of(dtype), (hipStream_t)stream);
    return check_launch("ldx_op_attention");
}
int ldx_op_xattn_block(void* H, int ldh, int64_t M, int N, int C, int heads, const float* ln_gamma, const float* ln_beta, float eps,
                       const void* Wq, const void* Wo, const float* bo, const void* K, int ldk, const void* V, int ldv, int Mk,
                       float scale, int dtype, void* stream) {
    XAttnArgs a{};
    a.H = H; a.ldh = ldh; a.M = (long)M; a.N = N; a.C = C; a.heads = heads; a.ln_g = ln_gamma; a.ln_b = ln_beta; a.eps = eps;
    a.Wq = Wq; a.Wo = Wo; a.bo = bo; a.K = K; a.ldk = ldk; a.V = V; a.ldv = ldv; a.Mk = Mk; a.scale = scale;
    if (!H || !ln_gamma || !ln_beta || !Wq || !Wo || !K || !V || M <= 0 || N <= 0 || ldh < C || !xattn_block_ok(a)) {
        set_error("ldx_op_xattn_block: shape not taken by the fused kernel (C = 320, 8 heads, Mk <= 80, N % 128 == 0, M % N == 0)"); return LDX_EINVAL; }
    launch_xattn_block(a, dtype_of(dtype), (hipStream_t)stream);
    return check_launch("ldx_op_xattn_block");
}
int ldx_op_ff_block(void* H, int ldh, int64_t M, int C, int inner, const float* ln_gamma, const float* ln_beta, float eps,
                    const void* W1, const float* b1, const void* W2, const float* b2, int dtype, void* stream) {
    FFBlockArgs a{};
    a.H = H; a.ldh = ldh; a.M = (long)M; a.C = C; a.inner = inner; a.ln_g = ln_gamma; a.ln_b = ln_beta; a.eps = eps; a.W1 = W1; a.b1 = b1; a.W2 = W2; a.b2 = b2;
    if (!H || !ln_gamma || !ln_beta || !W1 || !b1 || !W2 || M <= 0 || ldh < C || !ff_block_ok(a)) {
        set_error("ldx_op_ff_block: shape not taken by the fused kernel (C = 320, inner = 1280)"); return LDX_EINVAL; }
    launch_ff_block(a, dtype_of(dtype), (hipStream_t)stream);
    return check_launch("ldx_op_ff_block");
}
int ldx_op_rowgemm(const void* X, int ldx_, void* Y, int ldy, int64_t M, int N, int K, const void* W, const float* bias, const void* R, int ldr,
                   int pro, const float* gamma, const float* beta, float eps, const float* partial, int nchunk, int HW, int dtype, void* stream) {
    RowGemmArgs a{};
    a.X = X; a.ldx = ldx_; a.Y = Y; a.ldy = ldy; a.M = (long)M; a.N = N; a.K = K; a.W = W; a.bias = bias; a.R = R; a.ldr = ldr;
    a.pro = pro; a.g = gamma; a.b = beta; a.eps = eps; a.partial = partial; a.nchunk = nchunk; a.HW = HW; a.G = 32;
    if (!X || !Y || !W || ldx_ < K || ldy < N || !rowgemm_ok(a)) { set_error("ldx_op_rowgemm: shape not taken (K = 320 or 640, N a multiple of K, pro 0..2, GroupNorm: HW % (40960 / K) == 0, nchunk <= 256)"); return LDX_EINVAL; }
    launch_rowgemm(a, dtype_of(dtype), (hipStream_t)stream);
    return check_launch("ldx_op_rowgemm");
}
int ldx_op_attention_bias(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, void* O, int ldo, int B, int H, int Nq, int Mk, int D,
                          float scale, const float* bias, int bias_ld, int64_t bias_head_stride, int dtype, void* stream) {
    if (!Q || !K || !V || !O || !bias || D % 8 || D > 160 || D <= 0 || ldq % 8 || ldk % 8 || ldv % 8 || ldo % 4 || Mk <= 0 || bias_ld % 4 ||
        bias_ld < ((Mk + 63) / 64) * 64 || bias_head_stride % 4) {
        set_error("ldx_op_attention_bias: bad argument (bias_ld >= Mk rounded up to 64, multiples of 4)"); return LDX_EINVAL; }
    AttnArgs a{Q, ldq, K, ldk, V, ldv, O, ldo, B, H, Nq, Mk, D, scale, 0, bias, bias_ld, (long)bias_head_stride};
    launch_attention(a, dtype_of(dtype), (hipStream_t)stream);
    return check_launch("ldx_op_attention_bias");
}
int ldx_op_skinny(const float* x, int ldx_, const void* W, const float* bias, float* out, int ldo, int M, int N, int K, int in_act, int out_act, int dtype, void* stream) {
    if (!x || !W || !out || K % 8) { set_error("ldx_op_skinny: bad argument"); return LDX_EINVAL; }
    SkinnyArgs a{x, ldx_, W, bias, out, ldo, M, N, K, in_act, out_act};
    launch_skinny(a, dtype_of(dtype), (hipStream_t)stream);
    return check_launch("ldx_op_skinny");
}

}  // extern "C"
