// ldx UNet engine: host-side planner / executor behind the C ABI (include/ldx.h).
//
// Mirrors, for the accelerated path only, what the reference spreads over
//   UNetModel1.__init__/forward          src/NeuralNetwork/unet.py:208-770
//   ResBlock1 / Downsample1 / Upsample1  src/AutoEncoders/ResBlock.py:75-335
//   SpatialTransformer / BasicTransformerBlock / FeedForward   src/NeuralNetwork/transformer.py:19-377
//   CrossAttention                       src/Attention/Attention.py:53-124
//   BaseModel.apply_model                src/Model/ModelBase.py:72-133
// but as a static launch plan over one activation arena: weights are packed once into MFMA-friendly
// [N][K] 16-bit matrices (conv3x3 -> [Cout][ky][kx][Cin], q/k/v fused, GEGLU rows slab-interleaved),
// activations stay NHWC 16-bit, channel concat is a column offset, and every op is one of the HIP
// kernels in csrc/*.hip.  No CPU fallback exists: without a HIP device every call fails.
#include "engine.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <thread>

namespace ldx {

thread_local std::string g_last_error;
void set_error(const std::string& s) { g_last_error = s; }

#define HIP_OK(expr)                                                                         \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                    \
            return LDX_EHIP;                                                                 \
        }                                                                                    \
    } while (0)

// ---------------------------------------------------------------------------------------------
// 16-bit conversions on the host (round-to-nearest-even)
static inline float half_to_float(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000) << 16;
    uint32_t exp = (h >> 10) & 0x1f, man = h & 0x3ff, out;
    if (exp == 0) {
        if (man == 0) out = sign;
        else {
            exp = 127 - 15 + 1;
            while (!(man & 0x400)) { man <<= 1; --exp; }
            man &= 0x3ff;
            out = sign | (exp << 23) | (man << 13);
        }
    } else if (exp == 31) out = sign | 0x7f800000u | (man << 13);
    else out = sign | ((exp + 127 - 15) << 23) | (man << 13);
    float f; memcpy(&f, &out, 4); return f;
}
static inline uint16_t float_to_half(float f) {
    uint32_t x; memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000;
    x &= 0x7fffffff;
    if (x >= 0x7f800000) return (uint16_t)(sign | 0x7c00 | ((x > 0x7f800000) ? 0x200 : 0));
    if (x >= 0x477ff000) return (uint16_t)(sign | 0x7c00);                       // overflow -> inf
    if (x < 0x33000001) return (uint16_t)sign;                                  // underflow -> 0
    int exp = (int)(x >> 23) - 127 + 15;
    uint32_t man = x & 0x7fffff;
    if (exp <= 0) {                                                              // subnormal
        man |= 0x800000;
        const int shift = 14 - exp;
        uint32_t hm = man >> shift;
        const uint32_t rem = man & ((1u << shift) - 1), halfway = 1u << (shift - 1);
        if (rem > halfway || (rem == halfway && (hm & 1))) ++hm;
        return (uint16_t)(sign | hm);
    }
    uint32_t hm = man >> 13;
    const uint32_t rem = man & 0x1fff;
    uint32_t out = ((uint32_t)exp << 10) | hm;
    if (rem > 0x1000 || (rem == 0x1000 && (hm & 1))) ++out;
    return (uint16_t)(sign | out);
}
static inline float bf16_to_float(uint16_t h) { uint32_t x = (uint32_t)h << 16; float f; memcpy(&f, &x, 4); return f; }
static inline uint16_t float_to_bf16(float f) {
    uint32_t x; memcpy(&x, &f, 4);
    if ((x & 0x7fffffff) > 0x7f800000) return (uint16_t)((x >> 16) | 0x40);
    x += 0x7fff + ((x >> 16) & 1);
    return (uint16_t)(x >> 16);
}

float HostTensor::at(size_t i) const {
    switch (dtype) {
        case LDX_F32: return ((const float*)data.data())[i];
        case LDX_F16: return half_to_float(((const uint16_t*)data.data())[i]);
        default: return bf16_to_float(((const uint16_t*)data.data())[i]);
    }
}

static void parallel_for(size_t n, const std::function<void(size_t, size_t)>& fn) {
    unsigned nt = std::thread::hardware_concurrency();
    if (nt == 0) nt = 4;
    if (nt > 32) nt = 32;
    if (n < 1u << 16) { fn(0, n); return; }
    std::vector<std::thread> th;
    const size_t per = (n + nt - 1) / nt;
    for (unsigned t = 0; t < nt; ++t) {
        const size_t b = t * per, e = std::min(n, b + per);
        if (b >= e) break;
        th.emplace_back([=, &fn] { fn(b, e); });
    }
    for (auto& x : th) x.join();
}

// ---------------------------------------------------------------------------------------------
Engine::Engine(const ldx_unet_config& c, int dev) : cfg(c), device(dev) {
    dt = (c.compute_dtype == LDX_F16) ? DT_F16 : DT_BF16;
    cfg_share = getenv("LDX_CFG_SHARE") ? atoi(getenv("LDX_CFG_SHARE")) : 1;      // 0 off, 1 where it pays (share_for), 2 whenever possible
}
Engine::~Engine() {
    (void)hipSetDevice(device);
    for (void* p : dev_allocs) (void)hipFree(p);
    if (arena) (void)hipFree(arena);
    if (d_sigma_cfg) (void)hipFree(d_sigma_cfg);
    if (graph_exec) (void)hipGraphExecDestroy(graph_exec);
    for (PlanSnap& s : plan_cache) { if (s.arena) (void)hipFree(s.arena); if (s.graph_exec) (void)hipGraphExecDestroy(s.graph_exec); }
    for (hipEvent_t ev : prof_events) (void)hipEventDestroy(ev);
    if (cap_stream) (void)hipStreamDestroy(cap_stream);
}

int Engine::validate() const {
    auto bad = [&](const char* m) { set_error(std::string("unsupported UNet config: ") + m); return LDX_EINVAL; };
    if (cfg.model_channels <= 0 || cfg.model_channels % 64) return bad("model_channels must be a multiple of 64");
    if (cfg.context_dim <= 0 || cfg.context_dim % 64) return bad("context_dim must be a multiple of 64");
    if (cfg.num_levels < 1 || cfg.num_levels > 8) return bad("num_levels out of range");
    if (cfg.num_heads < 1) return bad("num_heads");
    if (cfg.in_channels < 1 || cfg.in_channels > 64 || cfg.out_channels < 1 || cfg.out_channels > 128) return bad("in/out channels");
    for (int l = 0; l < cfg.num_levels; ++l) {
        const int ch = cfg.model_channels * cfg.channel_mult[l];
        if (ch % cfg.num_heads || (ch / cfg.num_heads) % 8 || ch / cfg.num_heads > 160) return bad("head dim must be a multiple of 8 and <= 160");
        if (ch % 32) return bad("channels not divisible by 32 groups");
    }
    return LDX_OK;
}

int Engine::load_tensor(const char* key, const void* data, int dtype, const int64_t* shape, int ndim) {
    if (finalized) { set_error("ldx_load_tensor after ldx_finalize"); return LDX_ESTATE; }
    if (!key || !data || ndim < 0 || ndim > 8) { set_error("ldx_load_tensor: bad argument"); return LDX_EINVAL; }
    if (dtype != LDX_F32 && dtype != LDX_F16 && dtype != LDX_BF16) { set_error("ldx_load_tensor: bad dtype"); return LDX_EINVAL; }
    HostTensor t;
    t.dtype = dtype;
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) { t.shape.push_back(shape[i]); n *= (size_t)shape[i]; }
    t.numel = n;
    const size_t esz = dtype == LDX_F32 ? 4 : 2;
    t.data.resize(n * esz);
    memcpy(t.data.data(), data, n * esz);
    host[key] = std::move(t);
    return LDX_OK;
}

int Engine::set_tables(const float* ls, int n, const float* temb, int dim) {
    // after finalize the per-timestep emb_layers table (build_emb_table) and every cached plan depend on these tables
    if (finalized) { set_error("ldx_set_tables after ldx_finalize"); return LDX_ESTATE; }
    if (!ls || !temb || n <= 0 || dim != cfg.model_channels) { set_error("ldx_set_tables: bad argument (temb_dim must equal model_channels)"); return LDX_EINVAL; }
    HIP_OK(hipSetDevice(device));
    void* p = nullptr;
    HIP_OK(hipMalloc(&p, (size_t)n * 4)); dev_allocs.push_back(p);
    HIP_OK(hipMemcpy(p, ls, (size_t)n * 4, hipMemcpyHostToDevice));
    d_log_sigmas = (float*)p; n_sigmas = n;
    HIP_OK(hipMalloc(&p, (size_t)n * dim * 4)); dev_allocs.push_back(p);
    HIP_OK(hipMemcpy(p, temb, (size_t)n * dim * 4, hipMemcpyHostToDevice));
    d_temb = (float*)p;
    return LDX_OK;
}

const HostTensor* Engine::get(const std::string& key, std::initializer_list<int64_t> shape) {
    auto it = host.find(key);
    if (it == host.end()) { missing = key; return nullptr; }
    const HostTensor& t = it->second;
    if (shape.size()) {
        bool ok = t.shape.size() == shape.size();
        size_t i = 0;
        if (ok) for (int64_t s : shape) ok = ok && (t.shape[i++] == s);
        if (!ok) { missing = key + " (shape mismatch)"; return nullptr; }
    }
    return &t;
}

// upload a [rows][cols] matrix produced by getter(r, c) as 16-bit
void* Engine::upload16(size_t rows, size_t cols, const std::function<float(size_t, size_t)>& getter) {
    std::vector<uint16_t> buf(rows * cols);
    const bool bf = dt == DT_BF16;
    parallel_for(rows, [&](size_t b, size_t e) {
        for (size_t r = b; r < e; ++r)
            for (size_t c = 0; c < cols; ++c) {
                const float v = getter(r, c);
                buf[r * cols + c] = bf ? float_to_bf16(v) : float_to_half(v);
            }
    });
    void* p = nullptr;
    if (hipMalloc(&p, buf.size() * 2) != hipSuccess) return nullptr;
    dev_allocs.push_back(p);
    if (hipMemcpy(p, buf.data(), buf.size() * 2, hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    weight_bytes += buf.size() * 2;
    return p;
}
float* Engine::upload32(size_t n, const std::function<float(size_t)>& getter) {
    std::vector<float> buf(n);
    for (size_t i = 0; i < n; ++i) buf[i] = getter(i);
    void* p = nullptr;
    if (hipMalloc(&p, n * 4) != hipSuccess) return nullptr;
    dev_allocs.push_back(p);
    if (hipMemcpy(p, buf.data(), n * 4, hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    weight_bytes += n * 4;
    return (float*)p;
}

bool Engine::mk_linear(const std::string& pre, int N, int K, bool bias, LinearW& out, bool conv1x1) {
    const HostTensor* w = conv1x1 ? get(pre + ".weight", {N, K, 1, 1}) : get(pre + ".weight", {N, K});
    if (!w) return false;
    out.N = N; out.K = K;
    out.w = upload16(N, K, [&](size_t r, size_t c) { return w->at(r * K + c); });
    out.b = nullptr;
    if (bias) {
        const HostTensor* b = get(pre + ".bias", {N});
        if (!b) return false;
        out.b = upload32(N, [&](size_t i) { return b->at(i); });
        if (!out.b) return false;
    }
    return out.w != nullptr;
}
bool Engine::mk_conv3(const std::string& pre, int Cout, int Cin, int CinPad, LinearW& out) {
    const HostTensor* w = get(pre + ".weight", {Cout, Cin, 3, 3});
    const HostTensor* b = get(pre + ".bias", {Cout});
    if (!w || !b) return false;
    out.N = Cout; out.K = 9 * CinPad;
    // [Cout][ky][kx][CinPad]  <-  [Cout][Cin][ky][kx]
    out.w = upload16(Cout, (size_t)9 * CinPad, [&](size_t r, size_t c) {
        const size_t tap = c / CinPad, ci = c % CinPad;
        if ((int)ci >= Cin) return 0.f;
        return w->at((r * Cin + ci) * 9 + tap);
    });
    out.b = upload32(Cout, [&](size_t i) { return b->at(i); });
    return out.w && out.b;
}
bool Engine::mk_norm(const std::string& pre, int C, NormW& out) {
    const HostTensor* w = get(pre + ".weight", {C});
    const HostTensor* b = get(pre + ".bias", {C});
    if (!w || !b) return false;
    out.C = C;
    out.g = upload32(C, [&](size_t i) { return w->at(i); });
    out.b = upload32(C, [&](size_t i) { return b->at(i); });
    return out.g && out.b;
}

// Linear(LayerNorm(x)) with the norm folded in: y = rstd (x W'^T - mean c1) + c2 with W' = W .* gamma (per input column), c1[n] = sum_k W'[n][k]
// (of the values as STORED in 16 bit: the epilogue subtracts exactly what the MFMA accumulated for a constant row) and c2 = W beta + b.
bool Engine::mk_ln_folded(int N, int K, const std::function<float(size_t, size_t)>& W, const std::function<float(size_t)>& bias,
                          const std::string& norm_pre, LinearW& out, float*& c1) {
    const HostTensor* g = get(norm_pre + ".weight", {K});
    const HostTensor* be = get(norm_pre + ".bias", {K});
    if (!g || !be) return false;
    std::vector<float> gam(K), bet(K);
    for (int k = 0; k < K; ++k) { gam[k] = g->at(k); bet[k] = be->at(k); }
    out.N = N; out.K = K;
    out.w = upload16(N, K, [&](size_t r, size_t c) { return W(r, c) * gam[c]; });
    std::vector<float> v1(N), v2(N);
    const bool bf = dt == DT_BF16;
    parallel_for(N, [&](size_t b, size_t e) {
        for (size_t r = b; r < e; ++r) {
            double s1 = 0.0, s2 = 0.0;
            for (int k = 0; k < K; ++k) {
                const float w = W(r, k), wg = w * gam[k];
                s1 += bf ? bf16_to_float(float_to_bf16(wg)) : half_to_float(float_to_half(wg));
                s2 += (double)w * bet[k];
            }
            v1[r] = (float)s1; v2[r] = (float)(s2 + (bias ? bias(r) : 0.f));
        }
    });
    c1 = upload32(N, [&](size_t i) { return v1[i]; });
    out.b = upload32(N, [&](size_t i) { return v2[i]; });
    return out.w && c1 && out.b;
}

bool Engine::mk_res(const std::string& pre, int Cin, int Cout, ResW& r) {
    r.Cin = Cin; r.Cout = Cout;
    if (!mk_norm(pre + ".in_layers.0", Cin, r.gn1)) return false;
    if (!mk_conv3(pre + ".in_layers.2", Cout, Cin, Cin, r.conv1)) return false;
    // emb_layers.1 is batched with every other ResBlock's projection (one skinny GEMM per forward)
    const HostTensor* ew = get(pre + ".emb_layers.1.weight", {Cout, 4 * cfg.model_channels});
    const HostTensor* eb = get(pre + ".emb_layers.1.bias", {Cout});
    if (!ew || !eb) return false;
    r.emb_off = emb_total;
    emb_total += Cout;
    emb_srcs.push_back({ew, eb, Cout});
    if (!mk_norm(pre + ".out_layers.0", Cout, r.gn2)) return false;
    r.has_skip = Cin != Cout;
    const bool try_fuse = r.has_skip && Cin % 64 == 0 && !getenv("LDX_NO_FUSED_SKIP");
    if (!try_fuse && !mk_conv3(pre + ".out_layers.3", Cout, Cout, Cout, r.conv2)) return false;
    if (r.has_skip) {
        // out = conv2(h) + skip_connection(x): one implicit GEMM over K = 9*Cout + Cin (GemmArgs::A2), weights [Cout][ky][kx][Cout | Cin]
        const HostTensor* w2 = get(pre + ".out_layers.3.weight", {Cout, Cout, 3, 3});
        const HostTensor* b2 = get(pre + ".out_layers.3.bias", {Cout});
        const HostTensor* ws = get(pre + ".skip_connection.weight", {Cout, Cin, 1, 1});
        const HostTensor* bs = get(pre + ".skip_connection.bias", {Cout});
        if (try_fuse) {
            if (!w2 || !b2 || !ws || !bs) return false;
            r.fused_skip = true;
            r.conv2.N = Cout; r.conv2.K = 9 * Cout + Cin;
            r.conv2.w = upload16(Cout, (size_t)9 * Cout + Cin, [&](size_t rr, size_t c) {
                if (c >= (size_t)9 * Cout) return ws->at(rr * Cin + (c - (size_t)9 * Cout));
                const size_t tap = c / Cout, ci = c % Cout;
                return w2->at((rr * Cout + ci) * 9 + tap);
            });
            r.conv2.b = upload32(Cout, [&](size_t i) { return b2->at(i) + bs->at(i); });
            return r.conv2.w && r.conv2.b;
        }
        if (!mk_linear(pre + ".skip_connection", Cout, Cin, true, r.skip, true)) return false;
    }
    return true;
}

#ifndef LDX_LNFOLD_MAXROWS_DEFAULT
#define LDX_LNFOLD_MAXROWS_DEFAULT 8192      // rows of a transformer level up to which its LayerNorms are folded into the consuming GEMMs (emit_xf)
#endif
static inline bool q_prescale() { static const bool on = getenv("LDX_NO_QPRESCALE") == nullptr; return on; }

bool Engine::mk_xf(const std::string& pre, int C, int depth, XfW& x) {
    x.C = C; x.depth = depth;
    const int ctx = cfg.context_dim;
    if (!mk_norm(pre + ".norm", C, x.gn)) return false;
    if (!mk_linear(pre + ".proj_in", C, C, true, x.proj_in, true)) return false;
    if (!mk_linear(pre + ".proj_out", C, C, true, x.proj_out, true)) return false;
    x.blocks.resize(depth);
    for (int d = 0; d < depth; ++d) {
        XfBlockW& b = x.blocks[d];
        const std::string bp = pre + ".transformer_blocks." + std::to_string(d);
        if (!mk_norm(bp + ".norm1", C, b.ln1) || !mk_norm(bp + ".norm2", C, b.ln2) || !mk_norm(bp + ".norm3", C, b.ln3)) return false;
        // fused q|k|v for self attention
        const HostTensor* q = get(bp + ".attn1.to_q.weight", {C, C});
        const HostTensor* k = get(bp + ".attn1.to_k.weight", {C, C});
        const HostTensor* v = get(bp + ".attn1.to_v.weight", {C, C});
        if (!q || !k || !v) return false;
        // LDX_LNFOLD=1: fold norm1/2/3 into the q|k|v / q / GEGLU projections (48 launches and the normalised copies of h less per step).
        // Correct and tested, but measured neutral on MI355X (17.26 vs 17.14 ms per step): with every part of the fold switched off the
        // consuming GEMM is still 16 us slower at level 0 when it follows the GEMM that wrote h than when a LayerNorm launch sits between
        // them (50 -> 66 us; the statistics MFMAs, the LDS exchange and the epilogue add 8 more) - profiles/ubench/README.md.  Off by default.
        // Round 4: it does pay where the row-block kernels are not taken (512^2 step 6.01 -> 5.87 ms) and loses where they are (1024^2: 14.31 -> 15.04 with
        // the fold everywhere), so both copies of the three weights are kept (+0.3 GB for SD1.5) and the planner decides per input shape
        // (emit_xf: LDX_LNFOLD_MAXROWS); LDX_LNFOLD=0: plain weights only.
        static const bool fold_env = !(getenv("LDX_LNFOLD") != nullptr && atoi(getenv("LDX_LNFOLD")) == 0);
        b.ln_fold = fold_env;
        // softmax_scale * log2(e) folded into the q projections at load time (one rounding of c * Wq instead of rounding q and multiplying
        // every score): the attention ops then run with scale = 1 / log2(e), i.e. exp2(q.k - m) as before; LDX_NO_QPRESCALE=1 keeps the plain weights
        const float cq = q_prescale() ? (1.0f / std::sqrt((float)(C / cfg.num_heads))) * 1.44269504088896340736f : 1.0f;
        auto qkv_w = [&](size_t r, size_t c) {
            const HostTensor* s = r < (size_t)C ? q : (r < (size_t)2 * C ? k : v);
            return s->at((r % C) * C + c) * (r < (size_t)C ? cq : 1.0f);
        };
        const HostTensor* q2w = get(bp + ".attn2.to_q.weight", {C, C});
        if (!q2w) return false;
        if (b.ln_fold) {
            if (!mk_ln_folded(3 * C, C, qkv_w, nullptr, bp + ".norm1", b.qkv_f, b.c1_qkv)) return false;
            if (!mk_ln_folded(C, C, [&](size_t r, size_t c) { return q2w->at(r * C + c) * cq; }, nullptr, bp + ".norm2", b.q2_f, b.c1_q2)) return false;
        }
        {
            b.qkv.N = 3 * C; b.qkv.K = C; b.qkv.b = nullptr;
            b.qkv.w = upload16((size_t)3 * C, C, qkv_w);
            if (!b.qkv.w) return false;
            b.q2.N = C; b.q2.K = C; b.q2.b = nullptr;
            b.q2.w = upload16(C, C, [&](size_t r, size_t c) { return q2w->at(r * C + c) * cq; });
            if (!b.q2.w) return false;
        }
        if (!mk_linear(bp + ".attn1.to_out.0", C, C, true, b.o1)) return false;
        const HostTensor* k2 = get(bp + ".attn2.to_k.weight", {C, ctx});
        const HostTensor* v2 = get(bp + ".attn2.to_v.weight", {C, ctx});
        if (!k2 || !v2) return false;
        b.kv2.N = 2 * C; b.kv2.K = ctx; b.kv2.b = nullptr; b.kv2.w = nullptr;
        b.kv_off = kv_total;                       // columns [kv_off, kv_off + 2C) of the batched k|v projection
        kv_total += 2 * C;
        kv_srcs.push_back({k2, v2, C});
        if (!mk_linear(bp + ".attn2.to_out.0", C, C, true, b.o2)) return false;
        // GEGLU projection: rows permuted so each 64-column slab holds 32 value rows then their 32 gate rows
        const int inner = 4 * C;
        const HostTensor* fw = get(bp + ".ff.net.0.proj.weight", {2 * inner, C});
        const HostTensor* fb = get(bp + ".ff.net.0.proj.bias", {2 * inner});
        if (!fw || !fb) return false;
        auto src_row = [inner](size_t r) { const size_t slab = r / 64, within = r % 64; return within < 32 ? slab * 32 + within : inner + slab * 32 + (within - 32); };
        if (b.ln_fold) {
            if (!mk_ln_folded(2 * inner, C, [&](size_t r, size_t c) { return fw->at(src_row(r) * C + c); }, [&](size_t i) { return fb->at(src_row(i)); },
                              bp + ".norm3", b.ff1_f, b.c1_ff1)) return false;
        }
        {
            b.ff1.N = 2 * inner; b.ff1.K = C;
            b.ff1.w = upload16((size_t)2 * inner, C, [&](size_t r, size_t c) { return fw->at(src_row(r) * C + c); });
            b.ff1.b = upload32((size_t)2 * inner, [&](size_t i) { return fb->at(src_row(i)); });
            if (!b.ff1.w || !b.ff1.b) return false;
        }
        if (!mk_linear(bp + ".ff.net.2", C, inner, true, b.ff2)) return false;
    }
    return true;
}

int Engine::finalize() {
    if (finalized) return LDX_OK;
    if (kind != KIND_UNET) { set_error("finalize(): wrong engine kind"); return LDX_ESTATE; }
    int rc = validate();
    if (rc) return rc;
    HIP_OK(hipSetDevice(device));
    if (!d_log_sigmas) { set_error("ldx_finalize: call ldx_set_tables first"); return LDX_ESTATE; }
    const int mc = cfg.model_channels, ted = 4 * mc;
    bool ok = true;
    // --- structure walk, identical in order to UNetModel1.__init__ (unet.py:344-677) ---
    ok = ok && mk_linear("time_embed.0", ted, mc, true, te0) && mk_linear("time_embed.2", ted, ted, true, te2);
    ok = ok && mk_conv3("input_blocks.0.0", mc, cfg.in_channels, 64, conv_in);
    int ch = mc, td_i = 0, ib = 1;
    std::vector<int> chans{mc};
    for (int level = 0; ok && level < cfg.num_levels; ++level) {
        for (int nr = 0; ok && nr < cfg.num_res_blocks[level]; ++nr) {
            BlockW blk;
            const std::string pre = "input_blocks." + std::to_string(ib);
            const int co = cfg.channel_mult[level] * mc;
            blk.has_res = true;
            ok = ok && mk_res(pre + ".0", ch, co, blk.res);
            ch = co;
            const int depth = cfg.transformer_depth[td_i++];
            if (ok && depth > 0) { blk.has_xf = true; ok = mk_xf(pre + ".1", ch, depth, blk.xf); }
            in_blocks.push_back(std::move(blk)); chans.push_back(ch); ++ib;
        }
        if (ok && level != cfg.num_levels - 1) {
            BlockW blk; blk.has_down = true;
            ok = mk_conv3("input_blocks." + std::to_string(ib) + ".0.op", ch, ch, ch, blk.down);
            in_blocks.push_back(std::move(blk)); chans.push_back(ch); ++ib;
        }
    }
    if (ok && cfg.transformer_depth_middle >= -1) {
        has_middle = true;
        ok = mk_res("middle_block.0", ch, ch, mid_res0);
        if (ok && cfg.transformer_depth_middle >= 0) {
            mid_has_xf = true;
            ok = mk_xf("middle_block.1", ch, std::max(1, (int)cfg.transformer_depth_middle), mid_xf) && mk_res("middle_block.2", ch, ch, mid_res1);
            if (cfg.transformer_depth_middle == 0) { set_error("transformer_depth_middle == 0 unsupported"); return LDX_EINVAL; }
        }
    }
    int n_out = 0;
    for (int l = 0; l < cfg.num_levels; ++l) n_out += cfg.num_res_blocks[l] + 1;
    int td_o = n_out, ob = 0;
    for (int level = cfg.num_levels - 1; ok && level >= 0; --level) {
        for (int i = 0; ok && i <= cfg.num_res_blocks[level]; ++i) {
            BlockW blk;
            const std::string pre = "output_blocks." + std::to_string(ob);
            const int ich = chans.back(); chans.pop_back();
            const int co = mc * cfg.channel_mult[level];
            blk.has_res = true; blk.skip_ch = ich;
            ok = mk_res(pre + ".0", ch + ich, co, blk.res);
            ch = co;
            int sub = 1;
            const int depth = cfg.transformer_depth_output[--td_o];
            if (ok && depth > 0) { blk.has_xf = true; ok = mk_xf(pre + ".1", ch, depth, blk.xf); ++sub; }
            if (ok && level && i == cfg.num_res_blocks[level]) {
                blk.has_up = true;
                ok = mk_conv3(pre + "." + std::to_string(sub) + ".conv", ch, ch, ch, blk.up);
            }
            out_blocks.push_back(std::move(blk)); ++ob;
        }
    }
    ok = ok && mk_norm("out.0", ch, out_gn) && mk_conv3("out.2", cfg.out_channels, mc, mc, conv_out);
    if (ok) {
        // batched emb_layers: [emb_total][4*mc]
        std::vector<size_t> starts; size_t acc = 0;
        for (auto& s : emb_srcs) { starts.push_back(acc); acc += s.n; }
        auto find = [&](size_t r) { size_t i = std::upper_bound(starts.begin(), starts.end(), r) - starts.begin() - 1; return i; };
        emb_all.N = emb_total; emb_all.K = ted;
        emb_all.w = upload16(emb_total, ted, [&](size_t r, size_t c) { const size_t i = find(r); return emb_srcs[i].w->at((r - starts[i]) * ted + c); });
        emb_all.b = upload32(emb_total, [&](size_t r) { const size_t i = find(r); return emb_srcs[i].b->at(r - starts[i]); });
        ok = emb_all.w && emb_all.b;
    }
    if (ok && kv_total > 0) {
        std::vector<size_t> starts; size_t acc = 0;
        for (auto& s2 : kv_srcs) { starts.push_back(acc); acc += 2 * (size_t)s2.C; }
        const int ctxd = cfg.context_dim;
        kv_all.N = kv_total; kv_all.K = ctxd; kv_all.b = nullptr;
        kv_all.w = upload16(kv_total, ctxd, [&](size_t r, size_t c) {
            const size_t i = std::upper_bound(starts.begin(), starts.end(), r) - starts.begin() - 1;
            const size_t rr = r - starts[i]; const int C = kv_srcs[i].C;
            return (rr < (size_t)C ? kv_srcs[i].k : kv_srcs[i].v)->at((rr % C) * ctxd + c);
        });
        ok = kv_all.w != nullptr;
    }
    if (!ok) {
        if (!missing.empty()) { set_error("missing or mis-shaped weight: " + missing); return LDX_EMISSING; }
        set_error(std::string("weight upload failed: ") + hipGetErrorString(hipGetLastError()));
        return LDX_EHIP;
    }
    emb_srcs.clear(); kv_srcs.clear();
    host.clear();
    { const int rc2 = build_emb_table(); if (rc2) return rc2; }
    finalized = true;
    return LDX_OK;
}

// ---------------------------------------------------------------------------------------------
// arena allocator used at plan time (offsets), first-fit with coalescing
size_t Engine::a_alloc(size_t bytes) {
    bytes = (bytes + 255) & ~(size_t)255;
    for (size_t i = 0; i < free_list.size(); ++i) {
        if (free_list[i].second >= bytes) {
            const size_t off = free_list[i].first;
            if (free_list[i].second == bytes) free_list.erase(free_list.begin() + i);
            else { free_list[i].first += bytes; free_list[i].second -= bytes; }
            live[off] = bytes;
            return off;
        }
    }
    const size_t off = arena_top;
    arena_top += bytes;
    arena_peak = std::max(arena_peak, arena_top);
    live[off] = bytes;
    return off;
}
void Engine::a_free(size_t off) {
    auto it = live.find(off);
    if (it == live.end()) return;
    size_t sz = it->second;
    live.erase(it);
    free_list.emplace_back(off, sz);
    std::sort(free_list.begin(), free_list.end());
    for (size_t i = 0; i + 1 < free_list.size();) {
        if (free_list[i].first + free_list[i].second == free_list[i + 1].first) {
            free_list[i].second += free_list[i + 1].second;
            free_list.erase(free_list.begin() + i + 1);
        } else ++i;
    }
    if (!free_list.empty() && free_list.back().first + free_list.back().second == arena_top) {
        arena_top = free_list.back().first;
        free_list.pop_back();
    }
}

// Tile counters of the in-kernel split-K reduction (GemmArgs::sk_count): the engine's launches run in stream order and each leaves them at zero.
unsigned* Engine::sk_counters() {
#ifndef LDX_SK_FIXUP_BUILD
    return nullptr;                         // the in-kernel split-K reduction is not compiled in (gemm_sk_fixup() is constant false): no counters
#endif
    if (!d_sk_count) {
        if (hipMalloc((void**)&d_sk_count, sizeof(unsigned) * SK_COUNTERS) != hipSuccess) { d_sk_count = nullptr; return nullptr; }
        if (hipMemset(d_sk_count, 0, sizeof(unsigned) * SK_COUNTERS) != hipSuccess) { (void)hipFree(d_sk_count); d_sk_count = nullptr; return nullptr; }
        dev_allocs.push_back(d_sk_count);
    }
    return d_sk_count;
}

// Split-K workspace of one op: dead as soon as the op's reduce launch has run, so it is freed at once (the next buffer may reuse it).
size_t Engine::ws_alloc(size_t bytes) { const size_t off = a_alloc(bytes); a_free(off); return off; }

// ---- op emitters ---------------------------------------------------------------------------------
void Engine::op_gemm(const char* name, Act A, const LinearW& w, Act C, Act R, bool geglu, const float* rowvec, int rv_ld, int rpb) {
    Op o{}; o.kind = OP_GEMM; o.name = name;
    GemmArgs& g = o.g;
    g.A = ptr(A); g.lda = A.ld; g.W = w.w; g.M = A.rows; g.N = w.N; g.K = w.K; g.mode = 0;
    g.bias = w.b; g.rowvec = rowvec; g.rowvec_ld = rv_ld; g.rows_per_batch = rpb > 0 ? rpb : 1;
    g.geglu = geglu ? 1 : 0;
    g.R = R.valid ? ptr(R) : nullptr; g.ldr = R.ld;
    g.C = ptr(C); g.ldc = C.ld; g.Cf = nullptr;
    g.splitk = gemm_choose_splitk(g.M, g.N, g.K, geglu);
    if (g.splitk > 1) { g.ws = (float*)((uintptr_t)arena + ws_alloc(gemm_sk_ws_floats(g.M, g.N, g.splitk) * 4)); g.sk_count = sk_counters(); }
    o.flops = 2.0 * g.M * (double)g.N * g.K;
    o.bytes = 2.0 * ((double)g.M * g.K + (double)g.N * g.K + (double)g.M * (geglu ? g.N / 2 : g.N) * (R.valid ? 2 : 1));
    snprintf(o.klabel, sizeof(o.klabel), "gemm_kernel<%s,0>", dt == DT_BF16 ? "bf16" : "f16");
    ops.push_back(o);
    flops += o.flops;
}
void Engine::op_conv(const char* name, Act X, int B, int Hin, int Win, int Cin, const LinearW& w, int stride, int Hout, int Wout,
                     Act Y, Act R, const float* rowvec, int rv_ld, float* Cf, int ldcf) {
    Op o{}; o.kind = OP_GEMM; o.name = name;
    GemmArgs& g = o.g;
    g.A = ptr(X); g.lda = X.ld; g.W = w.w; g.M = B * Hout * Wout; g.N = w.N; g.K = w.K; g.mode = 1;
    g.Cin = Cin; g.Hin = Hin; g.Win = Win; g.Hout = Hout; g.Wout = Wout; g.stride = stride;
    if (stride == 1) { g.Hv = Hout; g.Wv = Wout; } else { g.Hv = Hin; g.Wv = Win; }
    g.resize = (g.Hv != Hin || g.Wv != Win) ? 1 : 0;
    g.bias = w.b; g.rowvec = rowvec; g.rowvec_ld = rv_ld; g.rows_per_batch = Hout * Wout;
    g.R = R.valid ? ptr(R) : nullptr; g.ldr = R.ld;
    g.C = Y.valid ? ptr(Y) : nullptr; g.ldc = Y.ld; g.Cf = Cf; g.ldcf = ldcf;
    g.splitk = gemm_choose_splitk(g.M, g.N, g.K, false);
    if (g.splitk > 1) { g.ws = (float*)((uintptr_t)arena + ws_alloc(gemm_sk_ws_floats(g.M, g.N, g.splitk) * 4)); g.sk_count = sk_counters(); }
    o.flops = 2.0 * g.M * (double)g.N * g.K;
    o.bytes = 2.0 * ((double)B * Hin * Win * Cin + (double)g.N * g.K + (double)g.M * g.N * (R.valid ? 2 : 1));
    snprintf(o.klabel, sizeof(o.klabel), "gemm_kernel<%s,1>", dt == DT_BF16 ? "bf16" : "f16");
    ops.push_back(o);
    flops += o.flops;
}
void Engine::op_gn(const char* name, Act X, Act Y, int B, int HW, const NormW& n, float eps, bool silu) {
    Op o{}; o.kind = OP_GN; o.name = name;
    GroupNormArgs& g = o.gn;
    g.X = ptr(X); g.ldx = X.ld; g.Y = ptr(Y); g.ldy = Y.ld; g.B = B; g.HW = HW; g.C = n.C; g.G = 32; g.eps = eps; g.silu = silu;
    g.gamma = n.g; g.beta = n.b; g.partial = (float*)(arena ? (char*)arena + gn_ws_off : nullptr);
    o.bytes = 2.0 * 3.0 * (double)B * HW * n.C;       // read (stats) + read + write (apply)
    snprintf(o.klabel, sizeof(o.klabel), "gn_stats+gn_apply");
    ops.push_back(o);
}
// SURVEY §8 a10-a11 / VERDICT r2 item 1c: GroupNorm statistics from the producing conv / GEMM epilogue.  A GroupNorm whose input tensor was written by
// the op right before it (so nothing else touches the shared statistics workspace in between) and whose producer launch can do it
// (gemm_gn_fuse: no split-K, tile width a multiple of the group width, tiles inside one image) drops its statistics pass — one full read
// of the activation and one launch.  The sums are still deterministic (fixed order per tile, then per tile row); what is given up is the
// bit-for-bit equality between different BATCH SIZES (the producer's tile shape, hence the summation order, depends on M).
void Engine::fuse_gn_stats() {
    for (size_t i = 1; i < ops.size(); ++i) {
        if (ops[i].kind == OP_GN && ops[i - 1].kind == OP_ROWGEMM) {          // row-block producer (proj_out + residual): statistics from its output stage
            RowGemmArgs& r = ops[i - 1].rg;
            GroupNormArgs& n = ops[i].gn;
            const int bm = 128 * 320 / r.K;
            static const bool off = getenv("LDX_GN_FUSE") && atoi(getenv("LDX_GN_FUSE")) == 0;
            if (off || r.pro == 2 || r.Y != n.X || r.ldy != n.ldx || r.N != n.C || r.N != r.K || n.G != 32 || (long)r.M != (long)n.B * n.HW || n.HW % bm || n.HW / bm > gn_ws_rows ||
                gn_uses_small_kernel(n.B, n.HW, n.C, n.G)) continue;      // last term: the one-launch small GroupNorm kernel takes it (norm.hip)
            r.gn_out = n.partial; r.gn_nchunk = n.HW / bm; r.HW = n.HW;
            n.stats_chunks = n.HW / bm;
            ops[i].bytes = 2.0 * 2.0 * (double)n.B * n.HW * n.C;
            snprintf(ops[i].klabel, sizeof(ops[i].klabel), "gn_apply(fused stats)");
            continue;
        }
        if (ops[i].kind != OP_GN || ops[i - 1].kind != OP_GEMM) continue;
        GemmArgs& g = ops[i - 1].g;
        GroupNormArgs& n = ops[i].gn;
        if (g.C != n.X || g.ldc != n.ldx || g.N != n.C || (long)g.M != (long)n.B * n.HW) continue;
        const int nchunk = gemm_gn_fuse(g, n.HW, n.G, gn_ws_rows);
        if (!nchunk) continue;
        g.gn_partial = n.partial;
        n.stats_chunks = nchunk;
        ops[i].bytes = 2.0 * 2.0 * (double)n.B * n.HW * n.C;       // read + write (apply only)
        snprintf(ops[i].klabel, sizeof(ops[i].klabel), "gn_apply(fused stats)");
    }
}
// Planner rule for the row-block kernels (rowgemm / xattn_block / ff_block): one workgroup per row block and one workgroup per CU, so below ~3/4 of the
// CUs the tile GEMMs win (SD1.5 512^2 has 64 row blocks per launch: step 6.14 -> 6.83 ms with the row-block kernels) and smaller problems stay on the
// separate launches.  LDX_ROWBLOCK_MINWG moves the limit (0: always).
// In a shared CFG prefix (half the batch: 128 row blocks at 1024^2, bs = 1) the limit is lower: the alternative there is the SAME number of rows on LayerNorm +
// tile-GEMM launches, which measured slower (LN 12.1 + GEMM 46.3 us against 40.6 for the row-block launch of the full batch; profiles/r06/share_*.txt).
static bool g_rowblock_prefix = false;          // set by emit_xf / fuse_gn_rowgemm while they emit prefix ops (planning is single-threaded per engine call)
static bool rowblock_fills_chip(long workgroups) {
    static const long min_wg = getenv("LDX_ROWBLOCK_MINWG") ? atol(getenv("LDX_ROWBLOCK_MINWG")) : 192;
    static const long min_wg_prefix = getenv("LDX_ROWBLOCK_MINWG_PREFIX") ? atol(getenv("LDX_ROWBLOCK_MINWG_PREFIX")) : 96;
    return workgroups >= (g_rowblock_prefix ? (min_wg_prefix < min_wg ? min_wg_prefix : min_wg) : min_wg);
}
// Row-block GEMM (rowgemm.hip) in place of [LayerNorm +] an N = 320 k, K = 320 projection; false (nothing emitted) when the kernel does not take it
bool Engine::op_rowgemm(const char* name, Act X, const LinearW& w, Act Y, Act R, int pro, const NormW* nw) {
    RowGemmArgs a{};
    a.X = ptr(X); a.ldx = X.ld; a.Y = ptr(Y); a.ldy = Y.ld; a.M = X.rows; a.N = w.N; a.K = w.K; a.W = w.w; a.bias = w.b;
    a.R = R.valid ? ptr(R) : nullptr; a.ldr = R.ld; a.pro = pro; a.eps = 1e-5f;
    if (nw) { a.g = nw->g; a.b = nw->b; }
    // Round 6: with nothing to fuse in front (pro = 0: to_out / proj_out + residual) the kernel only competes with the plain tile GEMM, whose output stage went
    // lean: at K = 640 and many rows (CFG batch 16: M = 65 536) the tile GEMM wins (92 against 156 us per launch); at bs = 1 (M = 8192) the two are level in the
    // step (13.27 against 13.28 ms) and the row block stays.
    static const long plain640_maxm = getenv("LDX_ROWGEMM_PLAIN640_MAXM") ? atol(getenv("LDX_ROWGEMM_PLAIN640_MAXM")) : 16384;
    if (pro == 0 && a.K == 640 && a.M > plain640_maxm) return false;
    if (!w.w || !rowgemm_ok(a) || !rowblock_fills_chip((a.M + 128 * 320 / a.K - 1) / (128 * 320 / a.K) * (a.K / 320))) return false;
    Op o{}; o.kind = OP_ROWGEMM; o.name = name; o.rg = a;
    o.flops = 2.0 * a.M * (double)a.N * a.K;
    o.bytes = 2.0 * ((double)a.M * a.K + (double)a.N * a.K + (double)a.M * a.N * (R.valid ? 2 : 1));
    snprintf(o.klabel, sizeof(o.klabel), "rowgemm<%s,%d>", dt == DT_BF16 ? "bf16" : "f16", pro);
    ops.push_back(o);
    flops += o.flops;
    return true;
}
// GroupNorm (no SiLU, C = 320, statistics already written by its producer) directly followed by the N = K = 320 GEMM that reads its output
// (SpatialTransformer norm + proj_in): the pair becomes one rowgemm launch with the GroupNorm apply as its prologue.  Runs after fuse_gn_stats().
void Engine::fuse_gn_rowgemm() {
    for (size_t i = 0; i + 1 < ops.size(); ++i) {
        g_rowblock_prefix = i < prefix_end;
        if (ops[i].kind != OP_GN || ops[i + 1].kind != OP_GEMM) continue;
        const GroupNormArgs& n = ops[i].gn;
        const GemmArgs& g = ops[i + 1].g;
        if (n.silu || n.stats_chunks <= 0 || n.stats_chunks > GN_NCHUNK || n.G != 32 || g.A != n.Y || g.lda != n.ldy || g.mode != 0 || g.K != n.C || g.geglu || g.ln_c1 ||
            g.f8 || g.C8 || g.Cf || g.rowvec || g.splitk > 1 || g.gate || g.R2 || g.act || g.oscale != 0.f || g.gn_partial || (long)g.M != (long)n.B * n.HW || !g.C) continue;
        RowGemmArgs a{};
        a.X = n.X; a.ldx = n.ldx; a.Y = g.C; a.ldy = g.ldc; a.M = g.M; a.N = g.N; a.K = g.K; a.W = g.W; a.bias = g.bias; a.R = g.R; a.ldr = g.ldr;
        a.pro = 2; a.g = n.gamma; a.b = n.beta; a.eps = n.eps; a.partial = n.partial; a.nchunk = n.stats_chunks; a.HW = n.HW; a.G = n.G;
        if (!rowgemm_ok(a) || !rowblock_fills_chip((a.M + 128 * 320 / a.K - 1) / (128 * 320 / a.K) * (a.K / 320))) continue;
        Op o{}; o.kind = OP_ROWGEMM; o.name = ops[i + 1].name; o.rg = a;
        o.flops = ops[i + 1].flops; o.bytes = ops[i + 1].bytes;
        snprintf(o.klabel, sizeof(o.klabel), "rowgemm<%s,2>", dt == DT_BF16 ? "bf16" : "f16");
        ops[i] = o;
        ops.erase(ops.begin() + (long)i + 1);
        if (i + 1 < prefix_end) --prefix_end;
    }
    g_rowblock_prefix = false;
}
void Engine::op_ln(const char* name, Act X, Act Y, const NormW& n) {
    Op o{}; o.kind = OP_LN; o.name = name;
    LayerNormArgs& l = o.ln;
    l.X = ptr(X); l.ldx = X.ld; l.Y = ptr(Y); l.ldy = Y.ld; l.rows = X.rows; l.C = n.C; l.eps = 1e-5f; l.gamma = n.g; l.beta = n.b;
    o.bytes = 2.0 * 2.0 * (double)X.rows * n.C;
    snprintf(o.klabel, sizeof(o.klabel), "ln_kernel");
    ops.push_back(o);
}
void Engine::op_attn(const char* name, const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, Act O, int B, int H, int Nq, int Mk, int D) {
    Op o{}; o.kind = OP_ATTN; o.name = name;
    AttnArgs& a = o.at;
    a.Q = Q; a.ldq = ldq; a.K = K; a.ldk = ldk; a.V = V; a.ldv = ldv; a.O = ptr(O); a.ldo = O.ld;
    a.B = B; a.H = H; a.Nq = Nq; a.Mk = Mk; a.D = D; a.scale = 1.0f / std::sqrt((float)D); a.causal = 0;
    if (attn_pipe_ok(a) || attn_pipe128_ok(a)) {       // key-block norms for the pipelined kernels' score bound (attn_pipe.hip): a few KiB, live for this op only
        const size_t off = a_alloc((size_t)B * H * ((Mk + 63) / 64) * 4); a.knorm_ws = (float*)((uintptr_t)arena + off); a_free(off);
    }
    if (attn512_ok(a)) {              // D = 512 (VAE): key splits when the query blocks alone leave CUs idle; fp32 partials live for this op only
        a.nsplit = attn512_splits(a);
        if (a.nsplit > 1) { const size_t off = a_alloc(attn512_ws_floats(a, a.nsplit) * 4); a.split_ws = (float*)((uintptr_t)arena + off); a_free(off); }
    }
    o.flops = 4.0 * B * H * (double)Nq * Mk * D;
    o.bytes = 2.0 * (double)B * H * D * (2.0 * Nq + 2.0 * Mk);
    {
        const int ks = D <= 32 ? 1 : D <= 64 ? 2 : D <= 96 ? 3 : D <= 128 ? 4 : 5;
        const int dtl = D / 16 + 1;
        const int cls = attention_dispatch_class(a);         // the device kernel's own name where one family takes the launch (rocprofv3 reports the same)
        if (attn512_ok(a)) snprintf(o.klabel, sizeof(o.klabel), "attn512_kernel<%s>x%d", dt == DT_BF16 ? "bf16" : "f16", a.nsplit);
        else if (cls == 1 || cls == 2) snprintf(o.klabel, sizeof(o.klabel), "%s<%s>%s", cls == 1 ? "attn40p_kernel" : "attn128p_kernel", dt == DT_BF16 ? "bf16" : "f16", Nq == Mk ? "self" : "cross");
        else snprintf(o.klabel, sizeof(o.klabel), "attn_kernel<%s,%d,%d>%s", dt == DT_BF16 ? "bf16" : "f16", ks, dtl, Nq == Mk ? "self" : "cross");
    }
    ops.push_back(o);
    flops += o.flops;
}

// rows [0, rows) of the view -> rows [rows, 2 * rows): the second half of a CFG batch takes over what was computed once for both (plan(): share)
void Engine::op_dup(const Act& a, int rows) {
    Op o{}; o.kind = OP_DUP; o.name = "cfg.dup";
    o.p0 = ptr(a); o.p1 = arena ? (char*)ptr(a) + (size_t)rows * a.ld * 2 : nullptr; o.i0 = rows; o.i1 = a.C; o.i2 = a.ld;
    o.bytes = 2.0 * 2.0 * (double)rows * a.C;
    snprintf(o.klabel, sizeof(o.klabel), "dup_rows");
    ops.push_back(o);
}

void Engine::dup_second_half(const DupReq& d) {
    static const bool copy_only = getenv("LDX_CFG_SHARE_COPY") && atoi(getenv("LDX_CFG_SHARE_COPY")) != 0;      // A/B switch: always the copy launch
    if (!copy_only && d.producer < ops.size()) {
        Op& o = ops[d.producer];
        if (o.kind == OP_GEMM && o.g.C == ptr(d.a) && o.g.ldc == d.a.ld && o.g.M == d.rows && !o.g.geglu && !o.g.C8) { o.g.dup_rows = d.rows; return; }
        if (o.kind == OP_ROWGEMM && o.rg.Y == ptr(d.a) && o.rg.ldy == d.a.ld && o.rg.M == d.rows) { o.rg.dup_rows = d.rows; return; }
    }
    op_dup(d.a, d.rows);
}

Act Engine::new_act(int rows, int C) {
    Act a; a.valid = true; a.owned = true; a.rows = rows; a.C = C; a.ld = C; a.col = 0;
    a.off = a_alloc((size_t)rows * C * 2);
    return a;
}
Act Engine::view(const Act& base, int col, int C) {
    Act a = base; a.owned = false; a.col = base.col + col; a.C = C; return a;
}
void Engine::release(const Act& a) { if (a.valid && a.owned) a_free(a.off); }

// ResBlock1._forward (ResBlock.py:315-335)
void Engine::emit_res(const ResW& r, Act X, Act OUT, int B, int H, int W) {
    const int M = B * H * W;
    Act t1 = new_act(M, r.Cin);
    op_gn("res.gn1", X, t1, B, H * W, r.gn1, r.eps, true);
    Act t2 = new_act(M, r.Cout);
    if (r.has_emb) op_conv("res.conv1", t1, B, H, W, r.Cin, r.conv1, 1, H, W, t2, Act{}, d_emb_all ? d_emb_all + r.emb_off : nullptr, emb_total);
    else op_conv("res.conv1", t1, B, H, W, r.Cin, r.conv1, 1, H, W, t2, Act{});
    release(t1);
    Act t3 = new_act(M, r.Cout);
    op_gn("res.gn2", t2, t3, B, H * W, r.gn2, r.eps, true);
    release(t2);
    if (r.has_skip && r.fused_skip) {
        op_conv("res.conv2+skip", t3, B, H, W, r.Cout, r.conv2, 1, H, W, OUT, Act{});
        GemmArgs& g = ops.back().g;
        g.A2 = ptr(X); g.lda2 = X.ld; g.Cin2 = r.Cin;
    } else if (r.has_skip) {
        Act t4 = new_act(M, r.Cout);
        op_gemm("res.skip", X, r.skip, t4, Act{});
        op_conv("res.conv2", t3, B, H, W, r.Cout, r.conv2, 1, H, W, OUT, t4);
        release(t4);
    } else {
        op_conv("res.conv2", t3, B, H, W, r.Cout, r.conv2, 1, H, W, OUT, X);
    }
    release(t3);
}

// SpatialTransformer.forward (transformer.py:342-377) + BasicTransformerBlock._forward (:186-245)
void Engine::emit_xf(const XfW& x, Act X, Act OUT, int B, int H, int W, Act ctx16, int Mc, int Bshare, const std::vector<DupReq>* dups) {
    const int M = B * H * W, C = x.C, heads = cfg.num_heads, D = C / heads;
    // Shared CFG prefix (plan(): share): up to the first cross-attention both halves of the batch hold the same values, so norm / proj_in / norm1 / q|k|v /
    // self-attention / to_out of the FIRST transformer block run on the first Bq samples (rows [0, Mq)) only
    int Bq = Bshare > 0 ? Bshare : B, Mq = Bq * H * W;
    g_rowblock_prefix = Bq != B;
    auto head = [&](const Act& a) { Act v = a; v.rows = Mq; v.owned = false; return v; };      // the first Mq rows of a buffer
    Act t1 = new_act(M, C);
    op_gn("xf.norm", X, t1, Bq, H * W, x.gn, 1e-6f, false);
    Act h = new_act(M, C);
    op_gemm("xf.proj_in", head(t1), x.proj_in, head(h), Act{});
    release(t1);
    // Folded LayerNorms (XfBlockW::ln_fold): the q|k|v / q / GEGLU GEMM reads h itself, accumulates each row's statistics from its own A
    // fragments and normalises in its epilogue (GemmArgs::ln_c1): no LayerNorm launch, no normalised copy of h.  A split-K consumer keeps a
    // plain (affine-free) LayerNorm launch in front of the folded weights.
    // Per level: folded where the row-block kernels do not take the level's projections (fewer than ~192 row blocks) and the level is small (measured,
    // same box: 512^2 step 6.00 -> 5.88 ms with all three levels folded, 430 -> 382 launches; 1024^2 14.23 -> 14.25 with only the 32^2 level folded
    // (neutral), 14.29 when the 64^2 level's rowgemm launches are replaced too; 8192 rows at C = 1280 (latent 256^2: HiresFix) lose 8 % of an evaluation:
    // the folded GEGLU projection runs on the 128-row tiles instead of the ping-pong ones).  The limit is on rows x C; LDX_LNFOLD_MAXROWS moves it
    // (rows at C = 320; 0: never).
    static const long fold_maxrows = getenv("LDX_LNFOLD_MAXROWS") ? atol(getenv("LDX_LNFOLD_MAXROWS")) : LDX_LNFOLD_MAXROWS_DEFAULT;
    const bool rowblocks = (C == 320 || C == 640) && rowblock_fills_chip(((long)M + 128 * 320 / C - 1) / (128 * 320 / C) * (C / 320));
    const bool fold = x.depth > 0 && x.blocks[0].ln_fold && !rowblocks && (long)M * C <= fold_maxrows * 320;      // 8192 rows at C = 320, 2048 at C = 1280
    Act n{};
    auto ln_gemm = [&](const char* ln_name, const char* name, const NormW& ln, const LinearW& w, const float* c1, Act Cc, bool geglu) {
        if (fold && gemm_choose_splitk(Mq, w.N, w.K, geglu) == 1) {
            op_gemm(name, head(h), w, head(Cc), Act{}, geglu);
            GemmArgs& g = ops.back().g;
            g.ln_c1 = c1; g.ln_eps = 1e-5f;
            return;
        }
        if (!n.valid) n = new_act(M, C);
        NormW plain = ln;
        if (fold) { plain.g = nullptr; plain.b = nullptr; }     // gamma / beta already live in the folded weights / bias
        op_ln(ln_name, head(h), head(n), plain);
        op_gemm(name, head(n), w, head(Cc), Act{}, geglu);
    };
    for (int d = 0; d < x.depth; ++d) {
        const XfBlockW& b = x.blocks[d];
        Act qkv = new_act(M, 3 * C);
        if (fold || b.qkv.b || !op_rowgemm("xf.ln1+qkv", head(h), b.qkv, head(qkv), Act{}, 1, &b.ln1))        // LayerNorm + q|k|v projection as one launch (C = 320)
            ln_gemm("xf.ln1", "xf.qkv", b.ln1, fold ? b.qkv_f : b.qkv, b.c1_qkv, qkv, false);
        Act a = new_act(M, C);
        const char* base = (const char*)ptr(qkv);
        op_attn("xf.attn1", base, 3 * C, base + (size_t)C * 2, 3 * C, base + (size_t)2 * C * 2, 3 * C, a, Bq, heads, H * W, H * W, D);
        if (q_prescale()) ops.back().at.scale = 1.0f / 1.44269504088896340736f;       // the q rows of the projection already carry scale * log2(e)
        release(qkv);
        if (!op_rowgemm("xf.o1", head(a), b.o1, head(h), head(h), 0, nullptr))
            op_gemm("xf.o1", head(a), b.o1, head(h), head(h));                   // x += attn1(norm1(x))   (in place)
        if (Bq != B) {
            // the context enters here: from now on the halves differ.  The second half's rows take over the shared results that full-batch ops still read.
            prefix_end = ops.size();                               // every op so far ran on one half of the batch
            if (dups) for (const DupReq& dq : *dups) dup_second_half(dq);
            dup_second_half(DupReq{h, Mq, ops.size() - 1});       // the op just emitted (to_out + residual) wrote h
            Bq = B; Mq = M; g_rowblock_prefix = false;
        }
        // k|v of the context come from the one batched projection emitted at the start of the forward
        const char* kvb = (const char*)((uintptr_t)arena + kv_all_off) + (size_t)b.kv_off * 2;
        XAttnArgs xa{};
        xa.H = ptr(h); xa.ldh = h.ld; xa.M = M; xa.N = H * W; xa.C = C; xa.heads = heads; xa.ln_g = b.ln2.g; xa.ln_b = b.ln2.b; xa.eps = 1e-5f;
        xa.Wq = b.q2.w; xa.Wo = b.o2.w; xa.bo = b.o2.b; xa.K = kvb; xa.ldk = kv_total; xa.V = kvb + (size_t)C * 2; xa.ldv = kv_total; xa.Mk = Mc;
        xa.scale = q_prescale() ? 1.0f / 1.44269504088896340736f : 1.0f / std::sqrt((float)D);
        if (!fold && !b.q2.b && xattn_block_ok(xa) && rowblock_fills_chip((M + 127) / 128)) {
            // LayerNorm + q projection + attention over the context + out projection + residual as ONE launch (xattn_block.hip)
            Op o{}; o.kind = OP_XATTN; o.name = "xf.xattn2"; o.xa = xa;
            o.flops = 2.0 * 2.0 * M * (double)C * C + 4.0 * B * heads * (double)(H * W) * Mc * D;
            o.bytes = 2.0 * 2.0 * (double)M * C;
            snprintf(o.klabel, sizeof(o.klabel), "xattn_block<%s>", dt == DT_BF16 ? "bf16" : "f16");
            ops.push_back(o);
            flops += o.flops;
            release(a);
        } else {
            Act q = new_act(M, C);
            static const bool x2 = !(getenv("LDX_ROWGEMM_X2") && atoi(getenv("LDX_ROWGEMM_X2")) == 0);      // A/B switch for the two uses below
            if (!x2 || fold || b.q2.b || !op_rowgemm("xf.ln2+q2", h, b.q2, q, Act{}, 1, &b.ln2))          // C = 640: LayerNorm + q projection as one row-block launch
                ln_gemm("xf.ln2", "xf.q2", b.ln2, fold ? b.q2_f : b.q2, b.c1_q2, q, false);
            op_attn("xf.attn2", ptr(q), C, kvb, kv_total, kvb + (size_t)C * 2, kv_total, a, B, heads, H * W, Mc, D);
            if (q_prescale()) ops.back().at.scale = 1.0f / 1.44269504088896340736f;       // the q rows of the projection already carry scale * log2(e)
            release(q);
            if (!x2 || !op_rowgemm("xf.o2", a, b.o2, h, h, 0, nullptr))
                op_gemm("xf.o2", a, b.o2, h, h);                   // x += attn2(norm2(x), ctx)
            release(a);
        }
        FFBlockArgs fa{};
        fa.H = ptr(h); fa.ldh = h.ld; fa.M = M; fa.C = C; fa.inner = 4 * C; fa.ln_g = b.ln3.g; fa.ln_b = b.ln3.b; fa.eps = 1e-5f;
        fa.W1 = b.ff1.w; fa.b1 = b.ff1.b; fa.W2 = b.ff2.w; fa.b2 = b.ff2.b;
        if (!fold && ff_block_ok(fa) && rowblock_fills_chip((M + 127) / 128)) {
            // LayerNorm + GEGLU projection + down projection + residual as ONE launch (ff_block.hip): the [M][4C] activation never leaves the CUs
            Op o{}; o.kind = OP_FFBLOCK; o.name = "xf.ffblock"; o.fb = fa;
            o.flops = 2.0 * M * (double)C * (8.0 * C) + 2.0 * M * (double)C * (4.0 * C);
            o.bytes = 2.0 * 2.0 * (double)M * C;
            snprintf(o.klabel, sizeof(o.klabel), "ff_block<%s>", dt == DT_BF16 ? "bf16" : "f16");
            ops.push_back(o);
            flops += o.flops;
        } else {
            Act f = new_act(M, 4 * C);
            ln_gemm("xf.ln3", "xf.ff1", b.ln3, fold ? b.ff1_f : b.ff1, b.c1_ff1, f, true);        // GEGLU
            op_gemm("xf.ff2", f, b.ff2, h, h);                     // x = ff(norm3(x)) + x
            release(f);
        }
    }
    if (n.valid) release(n);
    static const bool po = !(getenv("LDX_ROWGEMM_PO") && atoi(getenv("LDX_ROWGEMM_PO")) == 0);      // A/B switch
    if (!po || !op_rowgemm("xf.proj_out", h, x.proj_out, OUT, X, 0, nullptr))
        op_gemm("xf.proj_out", h, x.proj_out, OUT, X);         // + x_in
    release(h);
}

// The batch of the shared CFG prefix for an evaluation (0 = every op on the full batch): a CFG evaluation over ONE latent batch (xB > 0, B2 == 2 xB, no
// c_concat), a model with a cross-attention for the prefix to end at, and enough rows per half that the prefix ops still fill the chip — below
// LDX_CFG_SHARE_MINROWS (default 8192: 512^2 at bs = 1 has 4096 and measured 5.73 against 5.69 ms per step shared) nothing is shared.
int Engine::share_for(int B2, int h, int w, int xB, bool denoise, bool concat) const {
    static const long share_minrows = getenv("LDX_CFG_SHARE_MINROWS") ? atol(getenv("LDX_CFG_SHARE_MINROWS")) : 8192;
    if (!cfg_share || !denoise || concat || xB <= 0 || B2 != 2 * xB || (cfg_share == 1 && (long)xB * h * w < share_minrows)) return 0;
    for (auto& blk : in_blocks) if (blk.has_xf) return xB;
    return 0;
}

int Engine::plan(int B2, int h, int w, int Mc, int share) {
    // dry run (arena == nullptr) measures the peak; second run binds real pointers
    for (int pass = 0; pass < 2; ++pass) {
        prefix_end = 0; g_rowblock_prefix = false;
        ops.clear(); flops = 0; free_list.clear(); live.clear(); arena_top = 0; arena_peak = 0;
        if (pass == 1) {
            if (arena && arena_cap < arena_peak_dry) { HIP_OK(hipFree(arena)); arena = nullptr; }
            if (!arena) { HIP_OK(hipMalloc(&arena, arena_peak_dry)); arena_cap = arena_peak_dry; }
        }
        const bool bind = pass == 1;
        void* saved_arena = arena;
        if (!bind) arena = nullptr;
        const int mc = cfg.model_channels, ted = 4 * mc;
        // fixed small buffers
        const size_t o_temb = a_alloc((size_t)B2 * mc * 4), o_e1 = a_alloc((size_t)B2 * ted * 4), o_e2 = a_alloc((size_t)B2 * ted * 4);
        const size_t o_emb = a_alloc((size_t)B2 * emb_total * 4);
        gn_ws_off = a_alloc(gn_ws_bytes(B2, (long)h * w));
        const size_t o_eps = a_alloc((size_t)B2 * h * w * cfg.out_channels * 4);
        auto f32p = [&](size_t off) { return bind ? (float*)((char*)arena + off) : (float*)nullptr; };
        d_temb_out = f32p(o_temb); d_e1 = f32p(o_e1); d_e2 = f32p(o_e2); d_emb_all = f32p(o_emb); d_eps = f32p(o_eps);

        // ---- level geometry & concat buffers (so skips are written in place, no torch.cat copy) ----
        struct Skip { Act act; int H, W; };
        std::vector<int> Hs{h}, Ws{w};
        for (int l = 1; l < cfg.num_levels; ++l) { Hs.push_back((Hs.back() + 1) / 2); Ws.push_back((Ws.back() + 1) / 2); }
        // simulate the channel bookkeeping of the ctor to size the concat buffers of the output blocks
        std::vector<int> in_ch{mc}, in_lv{0};
        { int c = mc; for (int l = 0; l < cfg.num_levels; ++l) { for (int r = 0; r < cfg.num_res_blocks[l]; ++r) { c = mc * cfg.channel_mult[l]; in_ch.push_back(c); in_lv.push_back(l); }
              if (l != cfg.num_levels - 1) { in_ch.push_back(c); in_lv.push_back(l + 1); } } }
        const int n_skips = (int)in_ch.size();
        // output block k consumes skip (n_skips-1-k); its h channels:
        std::vector<int> out_hch(n_skips);
        { int c = in_ch.back(); int k = 0;
          for (int l = cfg.num_levels - 1; l >= 0; --l) for (int i = 0; i <= cfg.num_res_blocks[l]; ++i) { out_hch[k++] = c; c = mc * cfg.channel_mult[l]; } }
        std::vector<Act> cat(n_skips);
        for (int k = 0; k < n_skips; ++k) {
            const int s = n_skips - 1 - k, lv = in_lv[s];
            cat[k] = new_act(B2 * Hs[lv] * Ws[lv], out_hch[k] + in_ch[s]);
        }
        auto skip_view = [&](int s) { const int k = n_skips - 1 - s; return view(cat[k], out_hch[k], in_ch[s]); };

        // ---- ops ----
        Op po{}; po.kind = OP_PREP; po.name = "prep"; ops.push_back(po);          // pointers filled per call
        Act xin = new_act(B2 * h * w, 64);
        prep_xc_off = xin.off;
        Act ctx16 = new_act(B2 * Mc, cfg.context_dim);
        { Op o{}; o.kind = OP_CVT; o.name = "ctx.cvt"; o.cvt_out = ptr(ctx16); o.cvt_n = (size_t)B2 * Mc * cfg.context_dim; o.ctx_only = true; ops.push_back(o); }
        if (!d_emb_table) {       // per-step time-embedding MLP (LDX_EMB_TABLE=0); otherwise the prep kernel gathers the timestep's row of the table
            { Op o{}; o.kind = OP_SKINNY; o.name = "time_embed.0"; o.sk = SkinnyArgs{d_temb_out, mc, te0.w, te0.b, d_e1, ted, B2, ted, mc, 0, 1}; ops.push_back(o); }
            { Op o{}; o.kind = OP_SKINNY; o.name = "time_embed.2"; o.sk = SkinnyArgs{d_e1, ted, te2.w, te2.b, d_e2, ted, B2, ted, ted, 0, 1}; ops.push_back(o); }
            { Op o{}; o.kind = OP_SKINNY; o.name = "emb_layers"; o.sk = SkinnyArgs{d_e2, ted, emb_all.w, emb_all.b, d_emb_all, emb_total, B2, emb_total, ted, 0, 0}; ops.push_back(o); }
            flops += 2.0 * B2 * ((double)ted * mc + (double)ted * ted + (double)emb_total * ted);
        }
        Act kvall = new_act(B2 * Mc, kv_total);
        kv_all_off = kvall.off;
        op_gemm("xf.kv2_all", ctx16, kv_all, kvall, Act{});
        ops.back().ctx_only = true;

        int lv = 0, s = 0;
        Act hcur = skip_view(0);
        // Bp: batch of the ops in front of the first cross-attention (share > 0: one half of the CFG batch stands for both); `pend` = what they produce
        // that later full-batch ops read (the skip connections), duplicated into the second half's rows when the prefix ends (emit_xf)
        int Bp = share > 0 ? share : B2;
        std::vector<DupReq> pend;
        op_conv("conv_in", xin, Bp, h, w, 64, conv_in, 1, h, w, hcur, Act{});
        flops -= 2.0 * Bp * h * w * (double)mc * 9.0 * (64 - cfg.in_channels);   // padded channels are not algorithmic work
        release(xin);
        if (Bp != B2) pend.push_back({hcur, Bp * h * w, ops.size() - 1});
        for (auto& blk : in_blocks) {
            ++s;
            Act dst = skip_view(s);
            if (blk.has_down) {
                op_conv("down", hcur, Bp, Hs[lv], Ws[lv], in_ch[s - 1], blk.down, 2, Hs[lv + 1], Ws[lv + 1], dst, Act{});
                ++lv;
                if (Bp != B2) pend.push_back({dst, Bp * Hs[lv] * Ws[lv], ops.size() - 1});
            } else if (blk.has_xf) {
                Act mid = new_act(B2 * Hs[lv] * Ws[lv], blk.res.Cout);
                emit_res(blk.res, hcur, mid, Bp, Hs[lv], Ws[lv]);
                if (Bp != B2) {
                    pend.push_back({mid, Bp * Hs[lv] * Ws[lv], ops.size() - 1});          // proj_out's residual reads it for every sample (emit_res ends with the conv that writes it)
                    emit_xf(blk.xf, mid, dst, B2, Hs[lv], Ws[lv], ctx16, Mc, Bp, &pend);
                    Bp = B2; pend.clear();
                } else {
                    emit_xf(blk.xf, mid, dst, B2, Hs[lv], Ws[lv], ctx16, Mc);
                }
                release(mid);
            } else {
                emit_res(blk.res, hcur, dst, Bp, Hs[lv], Ws[lv]);
                if (Bp != B2) pend.push_back({dst, Bp * Hs[lv] * Ws[lv], ops.size() - 1});
            }
            hcur = dst;
        }
        // middle block (unet.py:537-584); its last op writes column 0 of the first concat buffer
        {
            const int Mm = B2 * Hs[lv] * Ws[lv];
            Act dst = view(cat[0], 0, out_hch[0]);
            if (!has_middle) { set_error("UNet config without a middle block is not supported by the planner"); arena = saved_arena; return LDX_EINVAL; }
            if (mid_has_xf) {
                Act m0 = new_act(Mm, mid_res0.Cout);
                emit_res(mid_res0, hcur, m0, B2, Hs[lv], Ws[lv]);
                Act m1 = new_act(Mm, mid_res0.Cout);
                emit_xf(mid_xf, m0, m1, B2, Hs[lv], Ws[lv], ctx16, Mc);
                release(m0);
                emit_res(mid_res1, m1, dst, B2, Hs[lv], Ws[lv]);
                release(m1);
            } else {
                emit_res(mid_res0, hcur, dst, B2, Hs[lv], Ws[lv]);
            }
        }
        // output blocks: input of block k is the whole concat buffer cat[k]
        int k = 0;
        Act final_h{};
        for (auto& blk : out_blocks) {
            const int sidx = n_skips - 1 - k; (void)sidx;
            const bool last = (k == n_skips - 1);
            const int Hc = Hs[lv], Wc = Ws[lv], Mrows = B2 * Hc * Wc;
            // where does this block's result go?  column 0 of the next concat buffer (at the next block's resolution)
            const int co = blk.res.Cout;
            int nsub = 1 + (blk.has_xf ? 1 : 0) + (blk.has_up ? 1 : 0);
            auto target = [&](bool is_final_sub, int rows) -> Act {
                if (is_final_sub && !last) return view(cat[k + 1], 0, out_hch[k + 1]);
                return new_act(rows, co);
            };
            int sub = 0;
            Act cur = cat[k];
            Act r_out = target(++sub == nsub, Mrows);
            emit_res(blk.res, cur, r_out, B2, Hc, Wc);
            release(cat[k]);
            cur = r_out;
            if (blk.has_xf) {
                Act x_out = target(++sub == nsub, Mrows);
                emit_xf(blk.xf, cur, x_out, B2, Hc, Wc, ctx16, Mc);
                release(cur);
                cur = x_out;
            }
            if (blk.has_up) {
                const int Hn = Hs[lv - 1], Wn = Ws[lv - 1];       // output_shape = hs[-1].shape (unet.py:754)
                Act u_out = target(++sub == nsub, B2 * Hn * Wn);
                op_conv("up", cur, B2, Hc, Wc, co, blk.up, 1, Hn, Wn, u_out, Act{});
                release(cur);
                cur = u_out; --lv;
            }
            final_h = cur;
            ++k;
        }
        // out: GroupNorm -> SiLU -> conv3x3 (unet.py:663-677), fp32 NHWC eps
        {
            const int Mrows = B2 * h * w;
            Act t = new_act(Mrows, mc);
            op_gn("out.gn", final_h, t, B2, h * w, out_gn, 1e-5f, true);
            release(final_h);
            op_conv("out.conv", t, B2, h, w, mc, conv_out, 1, h, w, Act{}, Act{}, nullptr, 0, d_eps, cfg.out_channels);
            release(t);
        }
        { Op o{}; o.kind = OP_FINISH; o.name = "finish"; ops.push_back(o); }
        release(ctx16); release(kvall);
        // what the shared prefix saves: every op in front of the first hand-over copy ran on half the batch (the context-only ops are not part of it)
        flops_shared = 0;
        if (share > 0) {
            for (size_t i = 0; i < prefix_end && i < ops.size(); ++i) if (!ops[i].ctx_only) flops_shared += ops[i].flops;
            flops_shared -= 2.0 * share * h * w * (double)mc * 9.0 * (64 - cfg.in_channels);        // conv_in's padded channels are not algorithmic work (see above)
        }
        fuse_gn_stats();
        fuse_gn_rowgemm();
        if (!bind) { arena_peak_dry = arena_peak; arena = saved_arena; }
    }
    pB2 = B2; ph = h; pw = w; pM = Mc; pShare = share;
    graph_valid = false; warm = false;
    kv_ptr = nullptr; kv_epoch = 0;          // a fresh arena holds no projected context
    return LDX_OK;
}

// The 22 emb_layers outputs for EVERY timestep of the table (ldx_set_tables), by the same three skinny launches a forward would run on its B2 rows:
// row t of d_emb_table is bit-identical to what those launches write for a sample whose timestep index is t (the skinny kernel's rows are independent).
int Engine::build_emb_table() {
    static const bool off = getenv("LDX_EMB_TABLE") && atoi(getenv("LDX_EMB_TABLE")) == 0;
    if (off || d_emb_table || !d_temb || n_sigmas <= 0 || emb_total <= 0 || emb_total % 4) return LDX_OK;
    const int mc = cfg.model_channels, ted = 4 * mc, n = n_sigmas;
    float *e1 = nullptr, *e2 = nullptr, *tab = nullptr;
    auto fail = [&](hipError_t err, const char* what) {       // nothing of a half-built table survives an error
        if (e1) (void)hipFree(e1);
        if (e2) (void)hipFree(e2);
        if (tab) (void)hipFree(tab);
        set_error(std::string(what) + ": " + hipGetErrorString(err));
        return LDX_EHIP;
    };
    hipError_t err;
    if ((err = hipMalloc((void**)&e1, (size_t)n * ted * 4)) != hipSuccess) return fail(err, "build_emb_table: hipMalloc");
    if ((err = hipMalloc((void**)&e2, (size_t)n * ted * 4)) != hipSuccess) return fail(err, "build_emb_table: hipMalloc");
    if ((err = hipMalloc((void**)&tab, (size_t)n * emb_total * 4)) != hipSuccess) return fail(err, "build_emb_table: hipMalloc");
    launch_skinny(SkinnyArgs{d_temb, mc, te0.w, te0.b, e1, ted, n, ted, mc, 0, 1}, dt, nullptr);
    launch_skinny(SkinnyArgs{e1, ted, te2.w, te2.b, e2, ted, n, ted, ted, 0, 1}, dt, nullptr);
    launch_skinny(SkinnyArgs{e2, ted, emb_all.w, emb_all.b, tab, emb_total, n, emb_total, ted, 0, 0}, dt, nullptr);
    if ((err = hipStreamSynchronize(nullptr)) != hipSuccess) return fail(err, "build_emb_table: hipStreamSynchronize");
    if ((err = hipGetLastError()) != hipSuccess) return fail(err, "build_emb_table: kernel launch");
    (void)hipFree(e1); (void)hipFree(e2);
    dev_allocs.push_back(tab);
    d_emb_table = tab;
    weight_bytes += (size_t)n * emb_total * 4;
    return LDX_OK;
}

double Engine::steady_flops() const {
    double f = flops;
    if (ctx_cache) for (const Op& o : ops) if (o.ctx_only) f -= o.flops;
    return f;
}
// the reference's arithmetic for the same evaluation: what runs plus what the shared CFG prefix computes once instead of twice
double Engine::algorithmic_flops() const { return steady_flops() + flops_shared; }

void Engine::plan_stash() {
    PlanSnap s;
    s.B2 = pB2; s.h = ph; s.w = pw; s.M = pM; s.share = pShare; s.ops = std::move(ops); s.flops = flops; s.flops_shared = flops_shared; s.arena = arena; s.arena_cap = arena_cap; s.arena_peak_dry = arena_peak_dry;
    s.gn_ws_off = gn_ws_off; s.prep_xc_off = prep_xc_off; s.kv_all_off = kv_all_off;
    s.d_temb_out = d_temb_out; s.d_e1 = d_e1; s.d_e2 = d_e2; s.d_emb_all = d_emb_all; s.d_eps = d_eps;
    s.kv_ptr = kv_ptr; s.kv_epoch = kv_epoch; s.kv_stream = kv_stream; s.g_ctxc = g_ctxc;
    s.graph_exec = graph_exec; s.graph_valid = graph_valid; s.warm = warm; s.g_x = g_x; s.g_s = g_s; s.g_ctx = g_ctx; s.g_out = g_out; s.g_den = g_den; s.g_xB = g_xB; s.g_cc = g_cc; s.g_ccn = g_ccn; s.g_t = g_t;
    s.fx_temb = fx_temb; s.fx_gemb = fx_gemb; s.fx_h1 = fx_h1; s.fx_vec = fx_vec; s.fx_svec = fx_svec; s.fx_mod = fx_mod; s.fx_tok = fx_tok;
    s.fb_s0 = fb_s0; s.fb_s1 = fb_s1; s.fb_x = fb_x; s.fb_first = fb_first; s.fb_res = fb_res; s.fb_part = fb_part;
    s.fb_B = fb_B; s.fb_L = fb_L; s.fb_Lt = fb_Lt; s.fb_C = fb_C; s.fb_a_end = fb_a_end; s.fb_b_end = fb_b_end;
    ops.clear(); arena = nullptr; arena_cap = 0; graph_exec = nullptr; graph_valid = false; warm = false; pB2 = ph = pw = pM = 0; pShare = 0;
    plan_cache.push_back(std::move(s));
    // every cached plan keeps its own arena resident: bound the cache by count (4) AND by bytes (LDX_PLAN_CACHE_GIB, default 16 GiB of
    // stashed arenas — a Flux plan per prompt length, a 2048^2 VAE plan of several GiB ...); oldest first, the newest entry always stays
    static const size_t cap_bytes = (size_t)((getenv("LDX_PLAN_CACHE_GIB") ? atof(getenv("LDX_PLAN_CACHE_GIB")) : 16.0) * (double)(1ull << 30));
    auto cached_bytes = [&]() { size_t b = 0; for (const PlanSnap& c : plan_cache) b += c.arena_cap; return b; };
    while (plan_cache.size() > 4 || (plan_cache.size() > 1 && cached_bytes() > cap_bytes)) {
        PlanSnap& o = plan_cache.front();
        if (o.arena) (void)hipFree(o.arena);
        if (o.graph_exec) (void)hipGraphExecDestroy(o.graph_exec);
        plan_cache.erase(plan_cache.begin());
    }
}
bool Engine::plan_restore(int B2, int h, int w, int Mc, int share) {
    for (size_t i = 0; i < plan_cache.size(); ++i) {
        PlanSnap& s = plan_cache[i];
        if (s.B2 != B2 || s.h != h || s.w != w || s.M != Mc || s.share != share) continue;
        ops = std::move(s.ops); flops = s.flops; flops_shared = s.flops_shared; arena = s.arena; arena_cap = s.arena_cap; arena_peak_dry = s.arena_peak_dry;
        gn_ws_off = s.gn_ws_off; prep_xc_off = s.prep_xc_off; kv_all_off = s.kv_all_off;
        d_temb_out = s.d_temb_out; d_e1 = s.d_e1; d_e2 = s.d_e2; d_emb_all = s.d_emb_all; d_eps = s.d_eps;
        kv_ptr = s.kv_ptr; kv_epoch = s.kv_epoch; kv_stream = s.kv_stream; g_ctxc = s.g_ctxc;
        graph_exec = s.graph_exec; graph_valid = s.graph_valid; warm = s.warm; g_x = s.g_x; g_s = s.g_s; g_ctx = s.g_ctx; g_out = s.g_out; g_den = s.g_den; g_xB = s.g_xB; g_cc = s.g_cc; g_ccn = s.g_ccn; g_t = s.g_t;
        fx_temb = s.fx_temb; fx_gemb = s.fx_gemb; fx_h1 = s.fx_h1; fx_vec = s.fx_vec; fx_svec = s.fx_svec; fx_mod = s.fx_mod; fx_tok = s.fx_tok;
        fb_s0 = s.fb_s0; fb_s1 = s.fb_s1; fb_x = s.fb_x; fb_first = s.fb_first; fb_res = s.fb_res; fb_part = s.fb_part;
        fb_B = s.fb_B; fb_L = s.fb_L; fb_Lt = s.fb_Lt; fb_C = s.fb_C; fb_a_end = s.fb_a_end; fb_b_end = s.fb_b_end;
        pB2 = B2; ph = h; pw = w; pM = Mc; pShare = share;
        plan_cache.erase(plan_cache.begin() + i);
        return true;
    }
    return false;
}

int Engine::exec_ops(hipStream_t ls, size_t op_begin, size_t op_end, int ctx_sel) {
    if (op_end > ops.size()) op_end = ops.size();
    const bool prof_now = profiling && !prof_graph;
    if (prof_now && prof_events.size() < 2 * ops.size()) {
        const size_t old = prof_events.size();
        prof_events.resize(2 * ops.size());
        for (size_t i = old; i < prof_events.size(); ++i) HIP_OK(hipEventCreate(&prof_events[i]));
    }
    for (size_t oi = op_begin; oi < op_end; ++oi) {
        const Op& o = ops[oi];
        if ((ctx_sel == 1 && o.ctx_only) || (ctx_sel == 2 && !o.ctx_only)) continue;
        if (prof_now) HIP_OK(hipEventRecord(prof_events[2 * oi], ls));
        switch (o.kind) {
            case OP_PREP: {
                PrepArgs p{};
                p.x = b_x; p.sigma = b_s; p.B = pB2; p.C = cfg.in_channels; p.H = ph; p.W = pw; p.Cpad = 64;
                p.xc = (char*)arena + prep_xc_off; p.log_sigmas = d_log_sigmas; p.n_sigmas = n_sigmas;
                p.temb_table = d_temb; p.temb_dim = cfg.model_channels; p.temb_out = d_temb_out; p.t_out = nullptr;
                p.scale_input = b_den ? 1 : 0; p.t_in = b_den ? b_t : b_s; p.xB = b_xB;      // b_t: indices from the caller (null: looked up from sigma on the device)
                p.cc = b_cc; p.Cx = cfg.in_channels - b_ccn;
                if (kind == KIND_UNET && d_emb_table) { p.emb_table = d_emb_table; p.emb_n = emb_total; p.emb_out = d_emb_all; }
                launch_prep(p, dt, ls);
            } break;
            case OP_CVT: launch_f32_to_t(b_ctx, o.cvt_out, o.cvt_n, dt, ls); break;
            case OP_SKINNY: launch_skinny(o.sk, dt, ls); break;
            case OP_GEMM: {
                launch_gemm(o.g, dt, ls);
                static const bool dbg_nan = getenv("LDX_DEBUG_NAN") != nullptr;          // debug: first GEMM whose 16-bit output holds a NaN / Inf
                if (dbg_nan && o.g.C && !prof_graph) {           // never inside a stream capture (the scan synchronises)
                    (void)hipStreamSynchronize(ls);
                    const int Nout = o.g.geglu ? o.g.N / 2 : o.g.N;
                    std::vector<uint16_t> hb((size_t)o.g.M * o.g.ldc);
                    (void)hipMemcpy(hb.data(), o.g.C, ((size_t)(o.g.M - 1) * o.g.ldc + Nout) * 2, hipMemcpyDeviceToHost);
                    long bad = 0, first = -1;
                    for (long m = 0; m < o.g.M; ++m) for (int n = 0; n < Nout; ++n) {
                        const uint16_t v = hb[(size_t)m * o.g.ldc + n];
                        const bool b = dt == DT_BF16 ? ((v & 0x7f80) == 0x7f80) : ((v & 0x7c00) == 0x7c00);
                        if (b) { ++bad; if (first < 0) first = m * (long)Nout + n; }
                    }
                    fprintf(stderr, "[nan] op %zu %-14s M%d N%d K%d sk%d ln%d rs%d: %ld bad%s\n", oi, o.name, o.g.M, o.g.N, o.g.K, o.g.splitk, o.g.ln_c1 ? 1 : 0,
                            0, bad, bad ? (std::string(" first at row ") + std::to_string(first / Nout) + " col " + std::to_string(first % Nout)).c_str() : "");
                }
            } break;
            case OP_MXQ: launch_mx_quant(o.mq, dt, ls); break;
            case OP_ATTN_MX: launch_attn_mx(o.am, dt, ls); break;
            case OP_MXVT: launch_mx_vt_quant(o.vt, dt, ls); break;
            case OP_GEMM2: launch_gemm2(o.g, o.g2, dt, ls); break;
            case OP_XATTN: launch_xattn_block(o.xa, dt, ls); break;
            case OP_FFBLOCK: launch_ff_block(o.fb, dt, ls); break;
            case OP_ROWGEMM: launch_rowgemm(o.rg, dt, ls); break;
            case OP_GN: launch_groupnorm(o.gn, dt, ls); break;
            case OP_LN: launch_layernorm(o.ln, dt, ls); break;
            case OP_ATTN:
                if (o.i3 == 1) { AttnArgs a = o.at; a.bias = b_bias; launch_attention(a, dt, ls); }      // per-call bias table (T5)
                else launch_attention(o.at, dt, ls);
                break;
            case OP_FINISH: {
                FinishArgs f{};
                f.eps = d_eps; f.ld = cfg.out_channels; f.x = b_den ? b_x : nullptr; f.sigma = b_s; f.out = b_out;
                f.B = pB2; f.C = cfg.out_channels; f.HW = ph * pw; f.xB = b_xB;
                launch_finish(f, ls);
            } break;
            case OP_VAEPREP: launch_vae_prep(b_x, o.p1, o.i0, o.i1, o.i2, o.i3, vae_pq, vae_pq ? vae_pq + o.i1 * o.i1 : nullptr, dt, ls); break;
            case OP_SOFTMAX: launch_softmax_rows(o.p1, o.i0, o.i1, o.i2, o.f0, dt, ls); break;
            case OP_CLAMP: launch_clamp01((const float*)o.p0, b_out, (size_t)o.i0, ls); break;
            case OP_EMBED: launch_clip_embed(b_ids, kind == KIND_T5 ? t5_tok : clip_tok, kind == KIND_T5 ? nullptr : clip_pos, o.p1, o.i0, o.i1, o.i2, o.i3,
                                              kind == KIND_T5 ? nullptr : clip_extra, kind == KIND_T5 ? 0 : clip_extra_n, dt, ls); break;
            case OP_CVT_OUT: if ((o.i3 ? b_out2 : b_out) != nullptr) launch_t_to_f32(o.p0, o.i3 ? b_out2 : b_out, (size_t)o.i0, dt, ls); break;
            case OP_PIXPREP: launch_pixels_prep(b_x, o.p1, o.i0, o.i1, o.i2, o.i3, o.f0, o.f1, dt, ls); break;
            case OP_COPY_OUT: HIP_OK(hipMemcpyAsync(b_out, o.p0, (size_t)o.cvt_n, hipMemcpyDeviceToDevice, ls)); break;
            case OP_MOMENTS: launch_mix_nhwc_to_nchw((const float*)o.p0, o.i1, b_out, o.i0, o.i1, o.i2, enc_qc, enc_qc ? enc_qc + o.i1 * o.i1 : nullptr, ls); break;
            case OP_DUP: launch_dup_rows(o.p0, o.p1, o.i0, o.i1, o.i2, dt, ls); break;
            case OP_FX_TEMB: launch_flux_temb(o.i0 ? b_guid : b_s, (float*)o.p1, pB2, 256, 1000.0f, ls); break;
            case OP_FX_SILU: launch_silu_f32((const float*)o.p0, (float*)o.p1, (size_t)o.i0, ls); break;
            case OP_FX_PATCH: launch_flux_patchify(b_x, o.p1, o.i0, o.i1, o.i2, o.i3, dt, ls); break;
            case OP_FX_CVT_CTX: launch_f32_to_t(b_ctx, o.cvt_out, o.cvt_n, dt, ls); break;
            case OP_FX_SKINNY_Y: { SkinnyArgs a = o.sk; a.x = b_y; launch_skinny(a, dt, ls); } break;
            case OP_FX_SKINNY_G: break;
            case OP_FX_ROPE: { QkRopeArgs a = o.rp; const int hp = a.D / 2; a.cosT = b_cos + (size_t)o.i0 * hp; a.sinT = b_sin + (size_t)o.i0 * hp;
                               if (a.Q8) launch_qk_norm_rope_mx(a, dt, ls); else launch_qk_norm_rope(a, dt, ls); } break;
            case OP_FX_UNPATCH: launch_flux_unpatchify(fx_tok, 4 * o.i1, b_den ? b_x : nullptr, b_s, b_out, o.i0, o.i1, o.i2, o.i3, ls); break;
        }
        if (prof_now) HIP_OK(hipEventRecord(prof_events[2 * oi + 1], ls));
    }
    if (prof_now) {
        HIP_OK(hipStreamSynchronize(ls));
        for (size_t i = op_begin; i < op_end; ++i) {
            float ms = 0.f;
            if ((ctx_sel == 1 && ops[i].ctx_only) || (ctx_sel == 2 && !ops[i].ctx_only)) continue;
            HIP_OK(hipEventElapsedTime(&ms, prof_events[2 * i], prof_events[2 * i + 1]));
            const Op& o = ops[i];
            std::string key = o.klabel[0] ? o.klabel : o.name;
            if (prof_detail) {
                char sh[96] = "";
                if (o.kind == OP_GEMM) snprintf(sh, sizeof(sh), " M%d N%d K%d sk%d%s", o.g.M, o.g.N, o.g.K, o.g.splitk, o.g.geglu ? " geglu" : "");
                else if (o.kind == OP_ATTN) snprintf(sh, sizeof(sh), " B%d H%d N%d M%d D%d", o.at.B, o.at.H, o.at.Nq, o.at.Mk, o.at.D);
                else if (o.kind == OP_ATTN_MX) snprintf(sh, sizeof(sh), " B%d H%d N%d M%d D128", o.am.B, o.am.H, o.am.Nq, o.am.Mk);
                else if (o.kind == OP_GN) snprintf(sh, sizeof(sh), " B%d HW%d C%d", o.gn.B, o.gn.HW, o.gn.C);
                else if (o.kind == OP_ROWGEMM) snprintf(sh, sizeof(sh), " M%ld N%d K%d", o.rg.M, o.rg.N, o.rg.K);
                else if (o.kind == OP_XATTN) snprintf(sh, sizeof(sh), " M%ld C%d", (long)o.xa.M, o.xa.C);
                else if (o.kind == OP_FFBLOCK) snprintf(sh, sizeof(sh), " M%ld C%d", (long)o.fb.M, o.fb.C);
                else if (o.kind == OP_LN) snprintf(sh, sizeof(sh), " R%d C%d", o.ln.rows, o.ln.C);
                else if (o.kind == OP_MXQ) snprintf(sh, sizeof(sh), " R%d K%d", o.mq.rows, o.mq.K);
                else if (o.kind == OP_GEMM2) snprintf(sh, sizeof(sh), " M%d+%d N%d K%d", o.g.M, o.g2.M, o.g.N, o.g.K);
                key += sh;
            }
            ProfEntry& pe = prof[key];
            pe.count += 1; pe.ms += ms; pe.flops += o.flops; pe.bytes += o.bytes;
        }
    }
    return LDX_OK;
}

int Engine::run_cfg(const float* x, float sigma, const float* ctx, int B, int h, int w, int Mc, float* out, hipStream_t st, int t_index) {
    if (B <= 0) { set_error("ldx_unet_denoise_cfg: bad argument"); return LDX_EINVAL; }
    if (t_index >= n_sigmas) { set_error("ldx_unet_denoise_cfg_t: t_index outside the sigma table"); return LDX_EINVAL; }
    HIP_OK(hipSetDevice(device));
    if (sigma_cfg_cap < 2 * B) {
        HIP_OK(hipStreamSynchronize(st));
        if (d_sigma_cfg) (void)hipFree(d_sigma_cfg);
        d_sigma_cfg = nullptr; sigma_cfg_cap = 0;
        HIP_OK(hipMalloc((void**)&d_sigma_cfg, sizeof(float) * 4 * B));
        sigma_cfg_cap = 2 * B;
    }
    float* d_t = d_sigma_cfg + sigma_cfg_cap;
    if (t_index >= 0) launch_fill2_f32(d_sigma_cfg, sigma, d_t, (float)t_index, 2 * B, st);          // outside the captured graph: the values change every step
    else launch_fill_f32(d_sigma_cfg, sigma, 2 * B, st);
    return run(x, d_sigma_cfg, ctx, 2 * B, h, w, Mc, out, true, st, B, nullptr, 0, t_index >= 0 ? d_t : nullptr);
}
int Engine::timestep_lookup(const float* sigma_dev, int n, int* out_dev, hipStream_t st) {
    if (kind != KIND_UNET || !d_log_sigmas) { set_error("ldx_unet_timestep: needs a UNet engine with its sigma table (ldx_set_tables)"); return LDX_ESTATE; }
    if (!sigma_dev || !out_dev || n < 0) { set_error("ldx_unet_timestep: bad argument"); return LDX_EINVAL; }
    HIP_OK(hipSetDevice(device));
    launch_timestep(sigma_dev, d_log_sigmas, n_sigmas, n, out_dev, st);
    HIP_OK(hipGetLastError());
    return LDX_OK;
}

int Engine::run(const float* x, const float* sigma_or_t, const float* ctx, int B2, int h, int w, int Mc, float* out, bool denoise, hipStream_t st, int xB,
                const float* c_concat, int cc_channels, const float* t_idx) {
    if (!denoise) t_idx = nullptr;            // forward mode: sigma_or_t already carries the indices
    if (c_concat && (cc_channels <= 0 || cc_channels >= cfg.in_channels || (denoise && cfg.in_channels - cc_channels != cfg.out_channels))) {
        set_error("ldx_unet_denoise_concat: c_concat channels must leave the latent's channels (in_channels - cc_channels == out_channels)"); return LDX_EINVAL;
    }
    if (!c_concat) cc_channels = 0;
    if (xB < 0 || (xB > 0 && B2 % xB != 0)) { set_error("ldx_unet_*: the batch of x must divide the evaluation batch"); return LDX_EINVAL; }
    if (!finalized) { set_error("ldx_unet_*: engine not finalized"); return LDX_ESTATE; }
    if (!x || !sigma_or_t || !ctx || !out || B2 <= 0 || h <= 0 || w <= 0 || Mc <= 0) { set_error("ldx_unet_*: bad argument"); return LDX_EINVAL; }
    HIP_OK(hipSetDevice(device));
    // CFG evaluation over one latent batch (ldx_unet_denoise_cfg*: xB > 0 and B2 == 2 xB): the two halves are identical up to the first cross-attention, which
    // the plan then computes once (plan(): share).  LDX_CFG_SHARE=0: every op on the full batch, as the concatenated ldx_unet_denoise call runs it.
    const int share = share_for(B2, h, w, xB, denoise, c_concat != nullptr);
    if (B2 != pB2 || h != ph || w != pw || Mc != pM || share != pShare) {
        HIP_OK(hipStreamSynchronize(st));
        if (pB2 > 0) plan_stash();
        if (!plan_restore(B2, h, w, Mc, share)) {
            int rc = plan(B2, h, w, Mc, share);
            if (rc) return rc;
        }
    }
    // context cache: the ctx_only ops run here, eagerly, only when this plan's buffers do not hold the projections of (ctx, epoch) yet; every
    // other op (and the captured graph) then leaves them out
    const bool ctxc = ctx_cache;
    b_ctx = ctx;
    if (ctxc && !(kv_ptr == ctx && kv_epoch == ctx_epoch && kv_stream == st)) {
        const bool pg = prof_graph; prof_graph = false;
        const int rc = exec_ops(st, 0, (size_t)-1, 2);
        prof_graph = pg;
        if (rc) return rc;
        kv_ptr = ctx; kv_epoch = ctx_epoch; kv_stream = st;
    }
    const bool same = (g_x == x && g_s == sigma_or_t && g_ctx == ctx && g_out == out && g_den == denoise && g_xB == xB && g_cc == c_concat && g_ccn == cc_channels && g_ctxc == ctxc && g_t == t_idx);
    // a captured graph has its pointers baked in: once a call arrives with other bindings (in ANY mode — the eager path below re-records
    // g_*), that graph must never be replayed against the new g_* (round 3: a stale graph was replayed after an eager call had moved g_*)
    if (!same) graph_valid = false;
    if (graph_mode && graph_valid && same) {
        HIP_OK(hipGraphLaunch(graph_exec, st));
        ++n_graph_replays;
        return LDX_OK;
    }
    // capture only once the same (plan, pointers) have been run eagerly before: the first eager pass
    // also performs the one-time hipFuncSetAttribute calls, which are illegal during capture.
    const bool use_graph = graph_mode && warm && same;
    g_x = x; g_s = sigma_or_t; g_ctx = ctx; g_out = out; g_den = denoise; g_xB = xB; g_cc = c_concat; g_ccn = cc_channels; g_ctxc = ctxc; g_t = t_idx; warm = true;
    hipStream_t ls = st;
    if (use_graph) {
        if (graph_exec) { (void)hipGraphExecDestroy(graph_exec); graph_exec = nullptr; }
        if (!cap_stream) HIP_OK(hipStreamCreateWithFlags(&cap_stream, hipStreamNonBlocking));
        HIP_OK(hipStreamSynchronize(st));
        HIP_OK(hipStreamBeginCapture(cap_stream, hipStreamCaptureModeThreadLocal));
        ls = cap_stream;
    }
    b_x = x; b_s = sigma_or_t; b_ctx = ctx; b_out = out; b_den = denoise; b_xB = xB; b_cc = c_concat; b_ccn = cc_channels; b_t = t_idx;
    prof_graph = use_graph;
    { int rc = exec_ops(ls, 0, (size_t)-1, ctxc ? 1 : 0); if (rc) return rc; }
    if (use_graph) {
        hipGraph_t g = nullptr;
        HIP_OK(hipStreamEndCapture(cap_stream, &g));
        HIP_OK(hipGraphInstantiate(&graph_exec, g, nullptr, nullptr, 0));
        (void)hipGraphDestroy(g);
        graph_valid = true;
        ++n_graph_captures;
        HIP_OK(hipGraphLaunch(graph_exec, st));
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error(std::string("kernel launch: ") + hipGetErrorString(e)); return LDX_EHIP; }
    return LDX_OK;
}

std::string Engine::profile_json() const {
    std::string s = "{";
    bool first = true;
    char buf[256];
    for (auto& kv : prof) {
        snprintf(buf, sizeof(buf), "%s\"%s\": {\"count\": %ld, \"ms\": %.6f, \"flops\": %.6e, \"bytes\": %.6e}", first ? "" : ", ",
                 kv.first.c_str(), kv.second.count, kv.second.ms, kv.second.flops, kv.second.bytes);
        s += buf; first = false;
    }
    return s + "}";
}

int64_t Engine::n_launches() const {
    int64_t n = 0;
    for (const Op& o : ops) {
        if (ctx_cache && o.ctx_only) continue;            // steady state of a sampling run: the context's projections are cached
        if (o.kind == OP_ATTN && o.at.nsplit > 1) { n += 2; continue; }      // split keys + merge launch (attn512.hip)
        if (o.kind == OP_GN) n += o.gn.stats_chunks > GN_NCHUNK ? 2 : (o.gn.stats_chunks > 0 ? 1 : 2);      // fold + apply / apply / statistics + apply
        else n += (o.kind == OP_GEMM && o.g.splitk > 1 && !gemm_sk_fixup(o.g)) ? 2 : 1;
    }
    return n;
}

}  // namespace ldx
