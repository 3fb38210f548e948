// T5 encoder stack (T5-XXL for Flux conditioning) as a static launch plan over the shared GEMM / attention / norm
// kernels.  Reference: src/clip/FluxClip.py — T5 :476-519, T5Stack :386-474, T5Block :325-384, T5LayerSelfAttention
// :268-323, T5Attention :72-266, T5LayerFF / T5DenseGatedActDense :15-70, T5LayerNorm :565-582.
//
// Per block:  x += o( attn( q(n), k(n), v(n) ; + relative-position bias ) ),  n = rms_norm(x)
//             x += wo( gelu_tanh(wi_0(m)) * wi_1(m) ),                          m = rms_norm(x)
// q|k|v are one fused GEMM; wi_0|wi_1 are one GEMM whose rows are slab-interleaved so that the gated activation is
// the GEMM epilogue (value = wi_1, gate = wi_0).  Attention is unscaled: the reference multiplies k by sqrt(d) to cancel
// SDPA's 1/sqrt(d) (:265-268).  The bias table comes from the host per call (same for every block, :316-321 / :453-456).
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

Engine::Engine(const ldx_t5_config& c, int dev) : cfg{}, device(dev) {
    kind = KIND_T5; tcfg = c;
    dt = (c.compute_dtype == LDX_F16) ? DT_F16 : DT_BF16;
}

int Engine::finalize_t5() {
    if (finalized) return LDX_OK;
    const ldx_t5_config& c = tcfg;
    auto bad = [&](const char* m) { set_error(std::string("unsupported T5 config: ") + m); return LDX_EINVAL; };
    if (c.d_model <= 0 || c.d_model % 64 || c.d_ff % 64 || c.d_model > 4096) return bad("d_model / d_ff must be multiples of 64, d_model <= 4096");
    if (c.num_heads <= 0 || c.d_model % c.num_heads || (c.d_model / c.num_heads) % 8 || c.d_model / c.num_heads > 160) return bad("head dim");
    if (c.num_layers <= 0 || c.vocab_size <= 0) return bad("num_layers / vocab_size");
    HIP_OK(hipSetDevice(device));
    const int E = c.d_model, F = c.d_ff;
    bool ok = true;
    const HostTensor* tok = get("shared.weight", {c.vocab_size, E});
    ok = tok != nullptr;
    if (ok) { t5_tok = upload32((size_t)c.vocab_size * E, [&](size_t i) { return tok->at(i); }); ok = t5_tok != nullptr; }
    auto rms = [&](const std::string& name, NormW& n) {
        const HostTensor* w = get(name + ".weight", {E});
        if (!w) return false;
        n.C = E; n.b = nullptr;
        n.g = upload32(E, [&](size_t i) { return w->at(i); });
        return n.g != nullptr;
    };
    t5_layers.resize(c.num_layers);
    for (int l = 0; ok && l < c.num_layers; ++l) {
        T5LayerW& L = t5_layers[l];
        const std::string a = "encoder.block." + std::to_string(l) + ".layer.0", f = "encoder.block." + std::to_string(l) + ".layer.1";
        ok = rms(a + ".layer_norm", L.ln1) && rms(f + ".layer_norm", L.ln2);
        const HostTensor *qw = get(a + ".SelfAttention.q.weight", {E, E}), *kw = get(a + ".SelfAttention.k.weight", {E, E}),
                         *vw = get(a + ".SelfAttention.v.weight", {E, E});
        ok = ok && qw && kw && vw;
        if (ok) {
            L.qkv.N = 3 * E; L.qkv.K = E; L.qkv.b = nullptr;
            L.qkv.w = upload16((size_t)3 * E, E, [&](size_t r, size_t cc) { const HostTensor* s = r < (size_t)E ? qw : (r < (size_t)2 * E ? kw : vw); return s->at((r % E) * E + cc); });
            ok = L.qkv.w != nullptr;
        }
        ok = ok && mk_linear(a + ".SelfAttention.o", E, E, false, L.o);
        const HostTensor *w0 = get(f + ".DenseReluDense.wi_0.weight", {F, E}), *w1 = get(f + ".DenseReluDense.wi_1.weight", {F, E});
        ok = ok && w0 && w1;
        if (ok) {
            // 64-row slabs: 32 value rows (wi_1) then their 32 gate rows (wi_0, through tanh-GELU)
            L.wi.N = 2 * F; L.wi.K = E; L.wi.b = nullptr;
            L.wi.w = upload16((size_t)2 * F, E, [&](size_t r, size_t cc) {
                const size_t slab = r / 64, within = r % 64;
                return within < 32 ? w1->at((slab * 32 + within) * E + cc) : w0->at((slab * 32 + within - 32) * E + cc); });
            ok = L.wi.w != nullptr;
        }
        ok = ok && mk_linear(f + ".DenseReluDense.wo", E, F, false, L.wo);
    }
    ok = ok && rms("encoder.final_layer_norm", t5_final_ln);
    if (!ok) {
        if (!missing.empty()) { set_error("missing or mis-shaped weight: " + missing); return LDX_EMISSING; }
        set_error(std::string("weight upload failed: ") + hipGetErrorString(hipGetLastError()));
        return LDX_EHIP;
    }
    host.clear();
    finalized = true;
    return LDX_OK;
}

int Engine::plan_t5(int B, int L) {
    const ldx_t5_config& c = tcfg;
    const int E = c.d_model, F = c.d_ff, M = B * L, heads = c.num_heads, D = E / heads;
    const int Lp = ((L + 63) / 64) * 64;
    for (int pass = 0; pass < 2; ++pass) {
        ops.clear(); flops = 0; free_list.clear(); live.clear(); arena_top = 0; arena_peak = 0;
        if (pass == 1) {
            if (arena && arena_cap < arena_peak_dry) { HIP_OK(hipFree(arena)); arena = nullptr; }
            if (!arena) { HIP_OK(hipMalloc(&arena, arena_peak_dry)); arena_cap = arena_peak_dry; }
        }
        void* saved = arena;
        if (pass == 0) arena = nullptr;
        Act x = new_act(M, E);
        { Op o{}; o.kind = OP_EMBED; o.name = "t5.embed"; o.p1 = ptr(x); o.i0 = B; o.i1 = L; o.i2 = E; o.i3 = c.vocab_size; ops.push_back(o); }
        Act n = new_act(M, E), qkv = new_act(M, 3 * E), a = new_act(M, E), f = new_act(M, F);
        auto rms = [&](const char* name, Act X, Act Y, const NormW& w) {
            op_ln(name, X, Y, w);
            ops.back().ln.eps = 1e-6f; ops.back().ln.rms = 1;
        };
        for (int l = 0; l < c.num_layers; ++l) {
            const T5LayerW& W = t5_layers[l];
            rms("t5.ln1", x, n, W.ln1);
            op_gemm("t5.qkv", n, W.qkv, qkv, Act{});
            const char* base = (const char*)ptr(qkv);
            op_attn("t5.attn", base, 3 * E, base + (size_t)E * 2, 3 * E, base + (size_t)2 * E * 2, 3 * E, a, B, heads, L, L, D);
            { Op& o = ops.back(); o.at.scale = 1.0f; o.at.bias_ld = Lp; o.at.bias_hs = (long)L * Lp; o.i3 = 1; }
            op_gemm("t5.o", a, W.o, x, x);                        // x += attention output
            rms("t5.ln2", x, n, W.ln2);
            op_gemm("t5.wi", n, W.wi, f, Act{}, true);            // gelu_tanh(wi_0 n) * (wi_1 n)
            ops.back().g.geglu = 2;
            op_gemm("t5.wo", f, W.wo, x, x);                      // x += FF output
        }
        rms("t5.final_ln", x, n, t5_final_ln);
        { Op o{}; o.kind = OP_CVT_OUT; o.name = "t5.out"; o.p0 = ptr(n); o.i0 = M * E; o.i3 = 0; ops.push_back(o); }
        if (pass == 0) { arena_peak_dry = arena_peak; arena = saved; }
    }
    pB2 = B; ph = L; pw = 0; pM = 0;
    return LDX_OK;
}

int Engine::run_t5(const int* ids, int B, int L, const float* bias, float* out, hipStream_t st) {
    if (!finalized || kind != KIND_T5) { set_error("ldx_t5_encode: not a finalized T5 engine"); return LDX_ESTATE; }
    if (!ids || !bias || !out || B <= 0 || L <= 0) { set_error("ldx_t5_encode: bad argument"); return LDX_EINVAL; }
    HIP_OK(hipSetDevice(device));
    if (B != pB2 || L != ph) {
        HIP_OK(hipStreamSynchronize(st));
        int rc = plan_t5(B, L);
        if (rc) return rc;
    }
    b_ids = ids; b_bias = bias; b_out = out; b_out2 = nullptr; prof_graph = false;
    int rc = exec_ops(st);
    if (rc) return rc;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error(std::string("kernel launch: ") + hipGetErrorString(e)); return LDX_EHIP; }
    return LDX_OK;
}

}  // namespace ldx
