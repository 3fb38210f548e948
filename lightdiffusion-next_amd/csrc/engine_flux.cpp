// Flux DiT on the shared planner / kernels.
//
//   Flux3.forward / forward_orig     src/BlackForest/Flux.py:658-778  (2x2 patchify, img/txt ids, blocks, LastLayer)
//   DoubleStreamBlock.forward        :298-348    SingleStreamBlock.forward :389-418    LastLayer.forward :455-471
//   Modulation :231-257, QKNorm/RMSNorm :148-200, attention + apply_rope :18-82, MLPEmbedder :105-131
//   timestep_embedding_flux          src/sample/sampling_util.py:78-104
//   CONST (flow) prediction          src/sample/sampling.py:100-155 (input unscaled, t = sigma, denoised = x - out*sigma)
//
// Layout: one joint token buffer X[B][Lt + Li][C] (txt rows first, then img rows) so the concat before the
// single-stream blocks (Flux.py:711) is free; every per-stream op addresses a row slice of it.  All 19*2 + 38 + 1
// adaLN modulation projections are one batched skinny GEMM on SiLU(vec) per forward.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "engine.h"

namespace ldx {

#define HIP_OK(expr)                                                                         \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                    \
            return LDX_EHIP;                                                                 \
        }                                                                                    \
    } while (0)

Engine::Engine(const ldx_flux_config& c, int dev) : cfg{}, device(dev) {
    kind = KIND_FLUX; fcfg = c;
    dt = (c.compute_dtype == LDX_F16) ? DT_F16 : DT_BF16;
}

int Engine::finalize_flux() {
    if (finalized) return LDX_OK;
    const ldx_flux_config& f = fcfg;
    auto bad = [&](const char* m) { set_error(std::string("unsupported Flux config: ") + m); return LDX_EINVAL; };
    const int C = f.hidden_size, H = f.num_heads;
    if (C <= 0 || C % 64 || f.mlp_hidden % 64 || C > 3072) return bad("hidden_size / mlp_hidden must be multiples of 64, hidden <= 3072");
    if (H <= 0 || C % H) return bad("hidden_size % num_heads");
    const int D = C / H;
    if (!(D == 16 || D == 32 || D == 64 || D == 128)) return bad("head dim must be 16, 32, 64 or 128");
    if (f.context_in_dim % 8 || f.vec_in_dim % 8 || (4 * f.in_channels) % 8) return bad("input dims must be multiples of 8");
    HIP_OK(hipSetDevice(device));
    bool ok = true;
    const int inC = 4 * f.in_channels;
    ok = ok && mk_linear("img_in", C, inC, true, fx_img_in) && mk_linear("txt_in", C, f.context_in_dim, true, fx_txt_in);
    ok = ok && mk_linear("time_in.in_layer", C, 256, true, fx_time0) && mk_linear("time_in.out_layer", C, C, true, fx_time1);
    ok = ok && mk_linear("vector_in.in_layer", C, f.vec_in_dim, true, fx_vec0) && mk_linear("vector_in.out_layer", C, C, true, fx_vec1);
    if (f.guidance_embed) ok = ok && mk_linear("guidance_in.in_layer", C, 256, true, fx_gd0) && mk_linear("guidance_in.out_layer", C, C, true, fx_gd1);
    auto scale_vec = [&](const std::string& key, float*& out) {
        const HostTensor* t = get(key, {D});
        if (!t) return false;
        out = upload32(D, [&](size_t i) { return t->at(i); });
        return out != nullptr;
    };
    auto add_mod = [&](const std::string& pre, int mult, int& off) {
        const HostTensor* w = get(pre + ".weight", {mult * C, C});
        const HostTensor* b = get(pre + ".bias", {mult * C});
        if (!w || !b) return false;
        off = fx_mod_total; fx_mod_total += mult * C;
        fx_mod_srcs.push_back({w, b, mult * C});
        return true;
    };
    auto mk_stream = [&](const std::string& p, const char* s, FluxStreamW& w) {
        const std::string a = p + "." + s;
        return add_mod(a + "_mod.lin", 6, w.mod_off) && mk_linear(a + "_attn.qkv", 3 * C, C, true, w.qkv) &&
               scale_vec(a + "_attn.norm.query_norm.scale", w.qs) && scale_vec(a + "_attn.norm.key_norm.scale", w.ks) &&
               mk_linear(a + "_attn.proj", C, C, true, w.proj) && mk_linear(a + "_mlp.0", f.mlp_hidden, C, true, w.mlp0) &&
               mk_linear(a + "_mlp.2", C, f.mlp_hidden, true, w.mlp2);
    };
    fx_double.resize(f.depth);
    for (int i = 0; ok && i < f.depth; ++i) {
        const std::string p = "double_blocks." + std::to_string(i);
        ok = mk_stream(p, "img", fx_double[i].img) && mk_stream(p, "txt", fx_double[i].txt);
    }
    fx_single.resize(f.depth_single);
    for (int i = 0; ok && i < f.depth_single; ++i) {
        const std::string p = "single_blocks." + std::to_string(i);
        FluxSingleW& s = fx_single[i];
        ok = add_mod(p + ".modulation.lin", 3, s.mod_off);
        // linear1 (C -> 3C + mlp) split into its qkv rows and its mlp rows (the mlp half gets the tanh-GELU epilogue)
        const HostTensor* w1 = ok ? get(p + ".linear1.weight", {3 * C + f.mlp_hidden, C}) : nullptr;
        const HostTensor* b1 = ok ? get(p + ".linear1.bias", {3 * C + f.mlp_hidden}) : nullptr;
        ok = ok && w1 && b1;
        if (ok) {
            s.lin1_qkv.N = 3 * C; s.lin1_qkv.K = C;
            s.lin1_qkv.w = upload16((size_t)3 * C, C, [&](size_t r, size_t c) { return w1->at(r * C + c); });
            s.lin1_qkv.b = upload32((size_t)3 * C, [&](size_t i2) { return b1->at(i2); });
            s.lin1_mlp.N = f.mlp_hidden; s.lin1_mlp.K = C;
            s.lin1_mlp.w = upload16((size_t)f.mlp_hidden, C, [&](size_t r, size_t c) { return w1->at((r + 3 * C) * C + c); });
            s.lin1_mlp.b = upload32((size_t)f.mlp_hidden, [&](size_t i2) { return b1->at(i2 + 3 * C); });
            ok = s.lin1_qkv.w && s.lin1_qkv.b && s.lin1_mlp.w && s.lin1_mlp.b;
        }
        ok = ok && mk_linear(p + ".linear2", C, C + f.mlp_hidden, true, s.lin2) &&
             scale_vec(p + ".norm.query_norm.scale", s.qs) && scale_vec(p + ".norm.key_norm.scale", s.ks);
    }
    ok = ok && add_mod("final_layer.adaLN_modulation.1", 2, fx_final_mod_off) && mk_linear("final_layer.linear", inC, C, true, fx_final);
    if (ok) {
        std::vector<size_t> starts; size_t acc = 0;
        for (auto& s : fx_mod_srcs) { starts.push_back(acc); acc += s.n; }
        auto find = [&](size_t r) { return (size_t)(std::upper_bound(starts.begin(), starts.end(), r) - starts.begin() - 1); };
        fx_mod_all.N = fx_mod_total; fx_mod_all.K = C;
        fx_mod_all.w = upload16(fx_mod_total, C, [&](size_t r, size_t c) { const size_t i = find(r); return fx_mod_srcs[i].w->at((r - starts[i]) * C + c); });
        fx_mod_all.b = upload32(fx_mod_total, [&](size_t r) { const size_t i = find(r); return fx_mod_srcs[i].b->at(r - starts[i]); });
        ok = fx_mod_all.w && fx_mod_all.b;
    }
    if (!ok) {
        if (!missing.empty()) { set_error("missing or mis-shaped weight: " + missing); return LDX_EMISSING; }
        set_error(std::string("weight upload failed: ") + hipGetErrorString(hipGetLastError()));
        return LDX_EHIP;
    }
    fx_mod_srcs.clear();
    host.clear();
    if (fx_fp8) {
        // MX copies of the block linears (the 16-bit copies stay: the tiny-M / ragged-K layers and the bf16 mode use them)
        if (C % 128 || f.mlp_hidden % 128) return bad("fp8 mode needs hidden_size and mlp_hidden to be multiples of 128");
        bool q = true;
        for (FluxDoubleW& d : fx_double)
            for (FluxStreamW* s : {&d.img, &d.txt}) q = q && mx_quantize_weight(s->qkv) && mx_quantize_weight(s->proj) && mx_quantize_weight(s->mlp0) && mx_quantize_weight(s->mlp2);
        for (FluxSingleW& s : fx_single) q = q && mx_quantize_weight(s.lin1_qkv) && mx_quantize_weight(s.lin1_mlp) && mx_quantize_weight(s.lin2);
        static const bool mod8 = !(getenv("LDX_FLUX_MOD_FP8") && atoi(getenv("LDX_FLUX_MOD_FP8")) == 0);      // round 6: the batched adaLN modulation projections too (6.4 -> 3.2 GB streamed per forward)
        if (mod8) q = q && mx_quantize_weight(fx_mod_all);
        if (!q || hipDeviceSynchronize() != hipSuccess) { set_error(std::string("MX weight quantisation failed: ") + hipGetErrorString(hipGetLastError())); return LDX_EHIP; }
    }
    finalized = true;
    return LDX_OK;
}

// W [N][K] 16-bit -> e4m3fn bytes [N][K] + E8M0 scales [K/128][N] (one per 32 consecutive k), on the device
bool Engine::mx_quantize_weight(LinearW& w) {
    void* w8 = nullptr; void* sw = nullptr;
    if (hipMalloc(&w8, (size_t)w.N * w.K) != hipSuccess) return false;
    dev_allocs.push_back(w8);
    if (hipMalloc(&sw, (size_t)(w.K / 128) * w.N * 4) != hipSuccess) return false;
    dev_allocs.push_back(sw);
    weight_bytes += (size_t)w.N * w.K + (size_t)(w.K / 128) * w.N * 4;
    MxQuantArgs a{w.w, w.K, w.N, w.K, w8, w.K, (uint32_t*)sw, w.N};
    launch_mx_quant(a, dt, nullptr);
    w.w8 = w8; w.sw = (uint32_t*)sw;
    return hipGetLastError() == hipSuccess;
}

int Engine::plan_flux(int B, int h, int w, int Lt) {
    const ldx_flux_config& f = fcfg;
    const int C = f.hidden_size, H = f.num_heads, D = C / H, MH = f.mlp_hidden;
    const int Li = (h / 2) * (w / 2), L = Lt + Li, inC = 4 * f.in_channels;
    for (int pass = 0; pass < 2; ++pass) {
        ops.clear(); flops = 0; free_list.clear(); live.clear(); arena_top = 0; arena_peak = 0;
        if (pass == 1) {
            if (arena && arena_cap < arena_peak_dry) { HIP_OK(hipFree(arena)); arena = nullptr; }
            if (!arena) { HIP_OK(hipMalloc(&arena, arena_peak_dry)); arena_cap = arena_peak_dry; }
        }
        void* saved = arena;
        if (pass == 0) arena = nullptr;
        auto f32buf = [&](size_t n) { const size_t off = a_alloc(n * 4); return (float*)((uintptr_t)arena + off); };
        fx_temb = f32buf((size_t)B * 256); fx_gemb = f32buf((size_t)B * 256); fx_h1 = f32buf((size_t)B * C);
        fx_vec = f32buf((size_t)B * C); fx_svec = f32buf((size_t)B * C); fx_mod = f32buf((size_t)B * fx_mod_total);
        fx_tok = f32buf((size_t)B * Li * inC);
        // first-block-cache state lives at the head of the arena and is never released: it survives from call to call
        fb_first = f32buf((size_t)B * Li * C); fb_res = f32buf((size_t)B * L * C); fb_part = f32buf(2 * 1024 + 8);
        { const size_t o0 = a_alloc((size_t)B * L * C * 2), o1 = a_alloc((size_t)B * L * C * 2); fb_s0 = (void*)((uintptr_t)arena + o0); fb_s1 = (void*)((uintptr_t)arena + o1); }
        auto skinny = [&](const char* name, OpKind kind, const float* x, int ldx_, const LinearW& lw, float* out, int out_act, int accum) {
            Op o{}; o.kind = kind; o.name = name; o.sk = SkinnyArgs{x, ldx_, lw.w, lw.b, out, lw.N, B, lw.N, lw.K, 0, out_act, accum}; ops.push_back(o);
            flops += 2.0 * B * (double)lw.N * lw.K;
        };
        // vec = time_in(temb(t)) + guidance_in(temb(g)) + vector_in(y)           (Flux.py:683-696)
        { Op o{}; o.kind = OP_FX_TEMB; o.name = "fx.temb_t"; o.p1 = fx_temb; o.i0 = 0; ops.push_back(o); }
        skinny("fx.time_in.0", OP_SKINNY, fx_temb, 256, fx_time0, fx_h1, 1, 0);
        skinny("fx.time_in.1", OP_SKINNY, fx_h1, C, fx_time1, fx_vec, 0, 0);
        if (f.guidance_embed) {
            { Op o{}; o.kind = OP_FX_TEMB; o.name = "fx.temb_g"; o.p1 = fx_gemb; o.i0 = 1; ops.push_back(o); }
            skinny("fx.guidance_in.0", OP_SKINNY, fx_gemb, 256, fx_gd0, fx_h1, 1, 0);
            skinny("fx.guidance_in.1", OP_SKINNY, fx_h1, C, fx_gd1, fx_vec, 0, 1);
        }
        skinny("fx.vector_in.0", OP_FX_SKINNY_Y, nullptr, f.vec_in_dim, fx_vec0, fx_h1, 1, 0);      // x bound per call (y)
        skinny("fx.vector_in.1", OP_SKINNY, fx_h1, C, fx_vec1, fx_vec, 0, 1);
        { Op o{}; o.kind = OP_FX_SILU; o.name = "fx.silu_vec"; o.p0 = fx_vec; o.p1 = fx_svec; o.i0 = B * C; ops.push_back(o); }
        skinny("fx.modulation_all", OP_SKINNY, fx_svec, C, fx_mod_all, fx_mod, 0, 0);
        if (fx_fp8 && fx_mod_all.w8) { SkinnyArgs& k = ops.back().sk; k.W8 = fx_mod_all.w8; k.SW = fx_mod_all.sw; k.sw_ld = fx_mod_all.N; }

        // joint token buffer and inputs
        Act X = new_act(B * L, C);
        Act ptok = new_act(B * Li, inC);
        { Op o{}; o.kind = OP_FX_PATCH; o.name = "fx.patchify"; o.p1 = ptr(ptok); o.i0 = B; o.i1 = f.in_channels; o.i2 = h; o.i3 = w; ops.push_back(o); }
        Act ctx16 = new_act(B * Lt, f.context_in_dim);
        { Op o{}; o.kind = OP_FX_CVT_CTX; o.name = "fx.ctx.cvt"; o.cvt_out = ptr(ctx16); o.cvt_n = (size_t)B * Lt * f.context_in_dim; ops.push_back(o); }
        auto rows = [&](const Act& t, int r0, int nr) { Act v = t; v.owned = false; v.off = t.off + (size_t)r0 * t.ld * 2; v.rows = nr; return v; };
        auto img_rows = [&](const Act& t, int b) { return rows(t, b * L + Lt, Li); };
        auto txt_rows = [&](const Act& t, int b) { return rows(t, b * L, Lt); };
        for (int b = 0; b < B; ++b) {
            op_gemm("fx.img_in", rows(ptok, b * Li, Li), fx_img_in, img_rows(X, b), Act{});
            op_gemm("fx.txt_in", rows(ctx16, b * Lt, Lt), fx_txt_in, txt_rows(X, b), Act{});
        }
        release(ptok); release(ctx16);

        // MX fp8 mode: an activation that feeds block linears gets an e4m3 shadow [B*L][K] + E8M0 scales [K/128][B*L], filled by
        // a quantise op in front of its consumers; the consumers then run the block-scaled MFMA GEMM on (shadow, MX weight).
        struct Q8 { char* y = nullptr; uint32_t* s = nullptr; int K = 0; };
        static const int fuse_mask = getenv("LDX_MX_FUSE") ? atoi(getenv("LDX_MX_FUSE")) : 7;      // experiment switch: 1 GEMM epilogue, 2 attention, 4 LayerNorm
        const bool fuse_gemm_q = fuse_mask & 1;
        const int RT = B * L;                                   // rows of every joint buffer = scale-array row stride
        auto new_q8 = [&](int K) { Q8 q; q.K = K; const size_t o8 = a_alloc((size_t)RT * K), os = a_alloc((size_t)(K / 128) * RT * 4);
                                   q.y = (char*)arena + o8; q.s = (uint32_t*)((char*)arena + os); return q; };
        auto row_of = [&](const Act& base, const Act& v) { return (int)((v.off - base.off) / ((size_t)base.ld * 2)); };
        // qo != null: the output goes to the MX shadow of the buffer Y lives in (returns true), not to Y
        auto ln_mod = [&](const char* name, Act Xin, Act Y, const float* shift, const float* scale, int rpb, const Q8* qo = nullptr, const Act* obase = nullptr) {
            Op o{}; o.kind = OP_LN; o.name = name;
            LayerNormArgs& l = o.ln;
            l.X = ptr(Xin); l.ldx = Xin.ld; l.Y = ptr(Y); l.ldy = Y.ld; l.rows = Xin.rows; l.C = C; l.eps = 1e-6f; l.gamma = nullptr; l.beta = nullptr;
            l.scale = scale; l.shift = shift; l.mod_ld = fx_mod_total; l.rows_per_batch = rpb;
            o.bytes = 2.0 * 2.0 * (double)Xin.rows * C; snprintf(o.klabel, sizeof(o.klabel), "ln_kernel");
            const bool fused = fx_fp8 && qo && (fuse_mask & 4);
            if (fused) { const int ro = row_of(*obase, Y); l.Y8 = qo->y + (size_t)ro * qo->K; l.ldy8 = qo->K; l.S8 = qo->s + ro; l.s8_ld = RT; }
            ops.push_back(o);
            return fused;
        };
        // MX fp8 attention (ldx_flux_set_fp8 mode 1, head dim 128; attn_mx.hip): the QKNorm + RoPE op writes q / k as MX fp8 (+ one scale dword per (row, head)) instead
        // of 16 bit, a transposing quantiser turns v into V^T bytes in the MFMA's key order, and the attention op runs both products on the block-scaled MFMA.
        static const bool attn8_env = !(getenv("LDX_FLUX_FP8_ATTN") && atoi(getenv("LDX_FLUX_FP8_ATTN")) == 0);      // A/B switch
        const bool attn8 = fx_fp8 && fx_fp8_attn && attn8_env && D == 128;
        const int Lp = (L + 127) / 128 * 128;
        char *a8_q = nullptr, *a8_k = nullptr, *a8_vt = nullptr; uint32_t *a8_sq = nullptr, *a8_sk = nullptr, *a8_sv = nullptr;
        if (attn8) {
            auto al = [&](size_t bytes) { return (char*)arena + a_alloc(bytes); };
            a8_q = al((size_t)RT * C); a8_k = al((size_t)RT * C); a8_vt = al((size_t)B * H * 128 * Lp);
            a8_sq = (uint32_t*)al((size_t)H * RT * 4); a8_sk = (uint32_t*)al((size_t)H * RT * 4); a8_sv = (uint32_t*)al((size_t)B * H * (Lp / 128) * 128 * 4);
        }
        auto rope = [&](const char* name, Act QKV, const float* qs, const float* ks, int tok0, const Act* qkv_base = nullptr) {
            Op o{}; o.kind = OP_FX_ROPE; o.name = name;
            o.rp = QkRopeArgs{ptr(QKV), QKV.ld, QKV.rows, L, H, D, qs, ks, nullptr, nullptr, 1e-6f};
            if (attn8 && qkv_base) {
                o.rp.Q8 = a8_q; o.rp.K8 = a8_k; o.rp.ld8 = C; o.rp.SQ = a8_sq; o.rp.SK = a8_sk; o.rp.s8_ld = RT; o.rp.row8 = row_of(*qkv_base, QKV);
                snprintf(o.klabel, sizeof(o.klabel), "qk_norm_rope_mx");
            }
            o.i0 = tok0;                                 // first token index of this slice in the pe tables
            ops.push_back(o);
        };
        auto vt_quant = [&](const char* name, Act QKVb, int b) {          // QKVb: the L rows of batch b
            Op o{}; o.kind = OP_MXVT; o.name = name;
            o.vt = MxVtArgs{(const char*)ptr(QKVb) + (size_t)2 * C * 2, QKVb.ld, 1, H, L, a8_vt + (size_t)b * H * 128 * Lp, a8_sv + (size_t)b * H * (Lp / 128) * 128, Lp};
            o.bytes = 3.0 * (double)L * C; snprintf(o.klabel, sizeof(o.klabel), "mx_vt_quant_kernel");
            ops.push_back(o);
        };
        auto quant = [&](const char* name, const Act& base, const Act& v, const Q8& q, int ncols = 0) {      // v: a row slice of base; its first ncols columns (0 = all)
            Op o{}; o.kind = OP_MXQ; o.name = name;
            const int r0 = row_of(base, v), K = ncols ? ncols : q.K;
            o.mq = MxQuantArgs{ptr(v), v.ld, v.rows, K, q.y + (size_t)r0 * q.K, q.K, q.s + r0, RT};
            o.bytes = 3.0 * (double)v.rows * K; snprintf(o.klabel, sizeof(o.klabel), "mx_quant_kernel");
            ops.push_back(o);
        };
        // linear on the rows of `v` (a row slice of `base`): 16-bit path, or MX path reading base's shadow q
        // qo != null (MX mode, no gate / residual): the output goes straight to the shadow qo of the buffer Cc lives in, at
        // Cc's rows and columns, quantised in the GEMM epilogue (Cc itself is not written)
        auto lin = [&](const char* name, const Act& base, const Act& v, const Q8& q, const LinearW& lw, Act Cc, Act R, const float* gate, int rpb, int act,
                       const Q8* qo = nullptr, const Act* obase = nullptr) {
            op_gemm(name, v, lw, Cc, R);
            GemmArgs& g = ops.back().g;
            g.gate = gate; g.gate_ld = fx_mod_total; g.rows_per_batch = rpb; g.act = act;
            if (gate && g.splitk > 1) g.splitk = 1;      // gate/act are not replicated in the split-K reduce path for safety
            if (fx_fp8 && lw.w8) {
                const int r0 = row_of(base, v);
                g.f8 = 1; g.A = q.y + (size_t)r0 * q.K; g.lda = q.K; g.SA = q.s + r0; g.sa_ld = RT; g.W = lw.w8; g.SW = lw.sw; g.sw_ld = lw.N;
                if (!gate) {
                    g.splitk = gemm_choose_splitk(g.M, g.N, g.K / 2, false);
                    if (g.splitk > 1) { g.ws = (float*)((uintptr_t)arena + ws_alloc(gemm_sk_ws_floats(g.M, g.N, g.splitk) * 4)); g.sk_count = sk_counters(); }
                }
                snprintf(ops.back().klabel, sizeof(ops.back().klabel), "gemm_kernel<mxfp8,0>");
                if (qo && fuse_gemm_q) {
                    const int ro = (int)((Cc.off - obase->off) / ((size_t)obase->ld * 2));
                    g.C = nullptr; g.C8 = qo->y + (size_t)ro * qo->K; g.ldc8 = qo->K; g.c8_col = Cc.col - obase->col; g.SC = qo->s + ro; g.sc_ld = RT; g.splitk = 1;
                }
            }
        };
        // merge the two independent plain GEMM ops just emitted into one two-problem launch
        static const bool group2 = !(getenv("LDX_FLUX_GROUP") && atoi(getenv("LDX_FLUX_GROUP")) == 0);      // experiment switch
        auto pair_last_two = [&](const char* name) {
            if (!group2 || ops.size() < 2) return;
            Op& A = ops[ops.size() - 2]; const Op& Bo = ops.back();
            if (A.kind != OP_GEMM || Bo.kind != OP_GEMM || A.g.mode || Bo.g.mode || A.g.geglu || Bo.g.geglu || A.g.f8 != Bo.g.f8) return;
            A.kind = OP_GEMM2; A.name = name; A.g2 = Bo.g; A.g.splitk = A.g2.splitk = 1; A.flops += Bo.flops; A.bytes += Bo.bytes;
            ops.pop_back();
        };
        // returns true if the attention kernel wrote the MX shadow qo of obase itself (head dim 128, large grid)
        auto attn = [&](const char* name, Act QKV, Act O, const Q8* qo = nullptr, const Act* obase = nullptr, int b = 0) {
            if (attn8) {
                Op o{}; o.kind = OP_ATTN_MX; o.name = name;
                AttnMxArgs& a = o.am;
                a.Q8 = a8_q + (size_t)b * L * C; a.ldq8 = C; a.SQ = a8_sq + (size_t)b * L; a.sq_ld = RT;
                a.K8 = a8_k + (size_t)b * L * C; a.ldk8 = C; a.SK = a8_sk + (size_t)b * L; a.sk_ld = RT;
                a.V8T = a8_vt + (size_t)b * H * 128 * Lp; a.SV = a8_sv + (size_t)b * H * (Lp / 128) * 128; a.Lp = Lp;
                a.B = 1; a.H = H; a.Nq = L; a.Mk = L; a.scale = 1.0f / std::sqrt((float)D);
                const bool fuse_out = qo && (fuse_mask & 2);
                if (fuse_out) { const int ro = row_of(*obase, O); a.O8 = qo->y + (size_t)ro * qo->K; a.ldo8 = qo->K; a.SO = qo->s + ro; a.so_ld = RT; }
                else { a.O = ptr(O); a.ldo = O.ld; }
                o.flops = 4.0 * H * (double)L * L * D; o.bytes = (double)H * D * (2.0 * L + 2.0 * L);
                snprintf(o.klabel, sizeof(o.klabel), "attn_mx_kernel");
                ops.push_back(o); flops += o.flops;
                return fuse_out;
            }
            const char* base = (const char*)ptr(QKV);
            op_attn(name, base, QKV.ld, base + (size_t)C * 2, QKV.ld, base + (size_t)2 * C * 2, QKV.ld, O, 1, H, QKV.rows, QKV.rows, D);
            AttnArgs& a = ops.back().at;
            if (!fx_fp8 || !qo || !(fuse_mask & 2) || !attention_mx_out_ok(a)) return false;
            const int ro = row_of(*obase, O);
            a.O8 = qo->y + (size_t)ro * qo->K; a.ldo8 = qo->K; a.SO = qo->s + ro; a.so_ld = RT;
            return true;
        };

        // ---- double-stream blocks ----
        Act QKV = new_act(B * L, 3 * C), AO = new_act(B * L, C), N1 = new_act(B * L, C), MLP = new_act(B * L, MH);
        Q8 qN1, qAO, qMLP, qCAT;
        if (fx_fp8) { qN1 = new_q8(C); qAO = new_q8(C); qMLP = new_q8(MH); qCAT = new_q8(C + MH); }
        fb_x = ptr(X); fb_B = B; fb_L = L; fb_Lt = Lt; fb_C = C;
        int blk_i = 0;
        for (const FluxDoubleW& blk : fx_double) {
            if (blk_i == 1) fb_a_end = ops.size();
            ++blk_i;
            for (int b = 0; b < B; ++b) {
                struct S { const FluxStreamW* w; Act x, n, qkv, ao, mlp; int rows; int tok0; };
                S st[2] = {{&blk.img, img_rows(X, b), img_rows(N1, b), img_rows(QKV, b), img_rows(AO, b), img_rows(MLP, b), Li, Lt},
                           {&blk.txt, txt_rows(X, b), txt_rows(N1, b), txt_rows(QKV, b), txt_rows(AO, b), txt_rows(MLP, b), Lt, 0}};
                // the image and the text stream run the same layer shapes on different weights: each pair of linears is one launch
                for (S& s : st) {
                    const float* m = fx_mod + (size_t)b * fx_mod_total;
                    const bool lq = ln_mod("fx.d.norm1", s.x, s.n, m + s.w->mod_off + 0 * C, m + s.w->mod_off + 1 * C, s.rows, &qN1, &N1);
                    if (fx_fp8 && !lq) quant("fx.d.q.norm1", N1, s.n, qN1);
                }
                for (S& s : st) lin("fx.d.qkv", N1, s.n, qN1, s.w->qkv, s.qkv, Act{}, nullptr, s.rows, 0);
                pair_last_two("fx.d.qkv x2");
                for (S& s : st) rope("fx.d.qknorm_rope", s.qkv, s.w->qs, s.w->ks, s.tok0, &QKV);
                if (attn8) vt_quant("fx.d.v.mx", rows(QKV, b * L, L), b);
                const bool aq = attn("fx.d.attn", rows(QKV, b * L, L), rows(AO, b * L, L), &qAO, &AO, b);          // joint [txt ; img] sequence
                if (fx_fp8 && !aq) quant("fx.d.q.attn", AO, rows(AO, b * L, L), qAO);
                for (S& s : st) {
                    const float* m = fx_mod + (size_t)b * fx_mod_total + s.w->mod_off;
                    lin("fx.d.proj", AO, s.ao, qAO, s.w->proj, s.x, s.x, m + 2 * C, s.rows, 0);        // x += gate1 * proj(attn)
                }
                pair_last_two("fx.d.proj x2");
                for (S& s : st) {
                    const float* m = fx_mod + (size_t)b * fx_mod_total + s.w->mod_off;
                    const bool lq = ln_mod("fx.d.norm2", s.x, s.n, m + 3 * C, m + 4 * C, s.rows, &qN1, &N1);
                    if (fx_fp8 && !lq) quant("fx.d.q.norm2", N1, s.n, qN1);
                }
                for (S& s : st) lin("fx.d.mlp0", N1, s.n, qN1, s.w->mlp0, s.mlp, Act{}, nullptr, s.rows, 2, &qMLP, &MLP);       // tanh-GELU
                pair_last_two("fx.d.mlp0 x2");
                if (fx_fp8 && !fuse_gemm_q) for (S& s : st) quant("fx.d.q.mlp", MLP, s.mlp, qMLP);
                for (S& s : st) {
                    const float* m = fx_mod + (size_t)b * fx_mod_total + s.w->mod_off;
                    lin("fx.d.mlp2", MLP, s.mlp, qMLP, s.w->mlp2, s.x, s.x, m + 5 * C, s.rows, 0);     // x += gate2 * mlp(...)
                }
                pair_last_two("fx.d.mlp2 x2");
            }
        }
        if (blk_i == 1) fb_a_end = ops.size();
        release(MLP);
        // ---- single-stream blocks on the joint sequence ----
        Act CAT = new_act(B * L, C + MH);                      // [attn | gelu(mlp)] : linear2's input (torch.cat, Flux.py:413)
        for (const FluxSingleW& blk : fx_single) {
            for (int b = 0; b < B; ++b) {
                const float* m = fx_mod + (size_t)b * fx_mod_total + blk.mod_off;
                Act xb = rows(X, b * L, L), nb = rows(N1, b * L, L), qb = rows(QKV, b * L, L), cb = rows(CAT, b * L, L);
                const bool lq = ln_mod("fx.s.pre_norm", xb, nb, m + 0 * C, m + 1 * C, L, &qN1, &N1);
                if (fx_fp8 && !lq) quant("fx.s.q.norm", N1, nb, qN1);
                lin("fx.s.lin1.qkv", N1, nb, qN1, blk.lin1_qkv, qb, Act{}, nullptr, L, 0);
                lin("fx.s.lin1.mlp", N1, nb, qN1, blk.lin1_mlp, view(cb, C, MH), Act{}, nullptr, L, 2, &qCAT, &CAT);
                pair_last_two("fx.s.lin1 x2");                    // linear1's two halves (different epilogues) share the rounds of one launch
                rope("fx.s.qknorm_rope", qb, blk.qs, blk.ks, 0, &QKV);
                if (attn8) vt_quant("fx.s.v.mx", qb, b);
                const bool aq = attn("fx.s.attn", qb, view(cb, 0, C), &qCAT, &CAT, b);
                if (fx_fp8 && !(aq && fuse_gemm_q)) quant("fx.s.q.cat", CAT, cb, qCAT, fuse_gemm_q ? C : 0);
                lin("fx.s.lin2", CAT, cb, qCAT, blk.lin2, xb, xb, m + 2 * C, L, 0);                 // x += gate * linear2(cat)
            }
        }
        release(CAT); release(QKV); release(AO);
        fb_b_end = ops.size();
        // ---- LastLayer on the img rows ----
        for (int b = 0; b < B; ++b) {
            const float* m = fx_mod + (size_t)b * fx_mod_total + fx_final_mod_off;
            ln_mod("fx.final.norm", img_rows(X, b), img_rows(N1, b), m + 0 * C, m + 1 * C, Li);       // chunk order: shift, scale
            op_gemm("fx.final.linear", img_rows(N1, b), fx_final, Act{}, Act{});
            GemmArgs& g = ops.back().g; g.C = nullptr; g.Cf = fx_tok + (size_t)b * Li * inC; g.ldcf = inC;
        }
        release(N1); release(X);
        { Op o{}; o.kind = OP_FX_UNPATCH; o.name = "fx.unpatchify"; o.i0 = B; o.i1 = f.in_channels; o.i2 = h; o.i3 = w; ops.push_back(o); }
        if (pass == 0) { arena_peak_dry = arena_peak; arena = saved; }
    }
    pB2 = B; ph = h; pw = w; pM = Lt;
    fb_reset();                                        // new shape: the cached residuals no longer apply (fbcache_nodes.py:56-66)
    return LDX_OK;
}

int Engine::run_flux(const float* x, const float* sigma, const float* ctx, const float* y, const float* guidance,
                     const float* pe_cos, const float* pe_sin, int B, int h, int w, int Lt, bool denoise, float* out, hipStream_t st) {
    if (!finalized || kind != KIND_FLUX) { set_error("ldx_flux_forward: not a finalized Flux engine"); return LDX_ESTATE; }
    if (!x || !sigma || !ctx || !y || !pe_cos || !pe_sin || !out || B <= 0 || h <= 0 || w <= 0 || Lt <= 0 || (h & 1) || (w & 1) ||
        (fcfg.guidance_embed && !guidance)) { set_error("ldx_flux_forward: bad argument (h, w must be even)"); return LDX_EINVAL; }
    HIP_OK(hipSetDevice(device));
    if (B != pB2 || h != ph || w != pw || Lt != pM) {
        HIP_OK(hipStreamSynchronize(st));
        // plans (op list + arena) are kept per shape like the UNet's: the multi-scale samplers alternate two resolutions and prompts
        // of different lengths change Lt; the first-block cache still resets on every shape change (fbcache_nodes.py:56-66)
        if (pB2 > 0) plan_stash();
        if (plan_restore(B, h, w, Lt)) fb_reset();
        else { int rc = plan_flux(B, h, w, Lt); if (rc) return rc; }
    }
    b_x = x; b_s = sigma; b_ctx = ctx; b_y = y; b_guid = guidance; b_cos = pe_cos; b_sin = pe_sin; b_out = out; b_den = denoise;
    prof_graph = false;
    int rc = LDX_OK;
    if (fb_threshold <= 0.f) {
        rc = exec_ops(st);
    } else {
        // ---- first-block cache (CachedTransformerBlocks.forward, first_block_cache.py:253-330) ----
        // ensure_cache_state (fbcache_nodes.py:56-75): reset unless the timestep strictly decreased; the reference reads
        // timestep[0].item() here, one 4-byte device read per forward
        float t0 = 0.f;
        HIP_OK(hipMemcpyAsync(&t0, sigma, sizeof(float), hipMemcpyDeviceToHost, st));
        HIP_OK(hipStreamSynchronize(st));
        if (!fb_prev_valid || t0 >= fb_prev_t) fb_reset();
        const size_t nj = (size_t)fb_B * fb_L * fb_C;
        // double block 0 with a snapshot of the joint stream in front of it (image rows: original_hidden_states)
        size_t b0_begin = fb_a_end;
        for (size_t i = 0; i < fb_a_end; ++i) if (std::string(ops[i].name) == "fx.d.norm1") { b0_begin = i; break; }
        rc = exec_ops(st, 0, b0_begin);
        if (rc) return rc;
        HIP_OK(hipMemcpyAsync(fb_s0, fb_x, nj * 2, hipMemcpyDeviceToDevice, st));
        rc = exec_ops(st, b0_begin, fb_a_end);
        if (rc) return rc;
        bool use = false;
        if (fb_have_first && fb_have_res) {          // get_can_use_cache -> are_two_tensors_similar (:105-148)
            launch_fb_diff(fb_x, fb_s0, fb_first, fb_B, fb_L, fb_Lt, fb_C, fb_part, fb_part + 2048, dt, st);
            float sums[2] = {0.f, 0.f};
            HIP_OK(hipMemcpyAsync(sums, fb_part + 2048, 2 * sizeof(float), hipMemcpyDeviceToHost, st));
            HIP_OK(hipStreamSynchronize(st));
            use = (sums[0] / sums[1]) < fb_threshold;      // mean|prev - cur| / mean|prev| (same element count)
        }
        if (use) {
            launch_fb_apply(fb_x, fb_res, nj, dt, st);                                   // hidden (+ encoder) states += cached residual
            ++fb_hits;
        } else {
            launch_fb_first(fb_x, fb_s0, fb_first, fb_B, fb_L, fb_Lt, fb_C, dt, st);    // set_buffer("first_hidden_states_residual")
            HIP_OK(hipMemcpyAsync(fb_s1, fb_x, nj * 2, hipMemcpyDeviceToDevice, st));
            rc = exec_ops(st, fb_a_end, fb_b_end);
            if (rc) return rc;
            launch_fb_residual(fb_x, fb_s1, fb_res, nj, dt, st);                         // final - after block 0, both streams
            fb_have_first = fb_have_res = true;
            ++fb_misses;
        }
        rc = exec_ops(st, fb_b_end, ops.size());
        fb_prev_t = t0; fb_prev_valid = true;                                            // update_cache_state
    }
    if (rc) { fb_reset(); return rc; }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { fb_reset(); set_error(std::string("kernel launch: ") + hipGetErrorString(e)); return LDX_EHIP; }
    return LDX_OK;
}

}  // namespace ldx
