// VAE decoder and CLIP text encoder on the same planner / kernels as the UNet (engine.cpp).
//
//   VAE   : Decoder.forward            src/AutoEncoders/VariationalAE.py:532-567
//           ResnetBlock.forward        src/AutoEncoders/ResBlock.py:383-406      (GroupNorm eps 1e-6, swish)
//           AttnBlock.forward          src/Attention/Attention.py:159-178        (1 head, D = C = 512)
//           Upsample.forward           src/AutoEncoders/VariationalAE.py:209-221 (nearest 2x + conv3x3)
//           AutoencodingEngine.decode  :130-145 (post_quant_conv) ; VAE.decode :690-722 (clamp, NHWC)
//   CLIP  : CLIPTextModel_.forward     src/clip/CLIPTextModel.py:51-107
//           CLIPLayer / CLIPAttention / CLIPMLP / CLIPEmbeddings   src/clip/Clip.py:14-294
#include <algorithm>
#include <cmath>
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

Engine::Engine(const ldx_vae_config& c, int dev) : cfg{}, device(dev) {
    kind = KIND_VAE; vcfg = c;
    dt = (c.compute_dtype == LDX_F16) ? DT_F16 : DT_BF16;
}
Engine::Engine(const ldx_clip_config& c, int dev) : cfg{}, device(dev) {
    kind = KIND_CLIP; ccfg = c;
    dt = (c.compute_dtype == LDX_F16) ? DT_F16 : DT_BF16;
}

// =============================================================================================
// VAE
bool Engine::mk_vae_res(const std::string& pre, int Cin, int Cout, ResW& r) {
    r.Cin = Cin; r.Cout = Cout; r.eps = 1e-6f; r.has_emb = false;
    if (!mk_norm(pre + ".norm1", Cin, r.gn1) || !mk_conv3(pre + ".conv1", Cout, Cin, Cin, r.conv1)) return false;
    if (!mk_norm(pre + ".norm2", Cout, r.gn2)) return false;
    r.has_skip = Cin != Cout;
    if (r.has_skip && Cin % 64 == 0 && !getenv("LDX_NO_FUSED_SKIP")) {
        // x = nin_shortcut(x); return x + h (ResBlock.py:383-406): folded into conv2 as a second K segment, like the UNet's ResBlock1
        const HostTensor* w2 = get(pre + ".conv2.weight", {Cout, Cout, 3, 3});
        const HostTensor* b2 = get(pre + ".conv2.bias", {Cout});
        const HostTensor* ws = get(pre + ".nin_shortcut.weight", {Cout, Cin, 1, 1});
        const HostTensor* bs = get(pre + ".nin_shortcut.bias", {Cout});
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
    if (!mk_conv3(pre + ".conv2", Cout, Cout, Cout, r.conv2)) return false;
    if (r.has_skip && !mk_linear(pre + ".nin_shortcut", Cout, Cin, true, r.skip, true)) return false;
    return true;
}

bool Engine::mk_vae_attn(const std::string& pre, int C, VaeAttnW& a) {
    if (!mk_norm(pre + ".norm", C, a.norm) || !mk_linear(pre + ".q", C, C, true, a.q, true) || !mk_linear(pre + ".k", C, C, true, a.k, true)) return false;
    // v is used as the A operand (V^T = Wv . h^T); its bias is folded through the softmax (rows sum to 1)
    // into the output projection's bias: b' = Wp . bv + bp.
    if (!mk_linear(pre + ".v", C, C, false, a.v, true)) return false;
    const HostTensor* wp = get(pre + ".proj_out.weight", {C, C, 1, 1});
    const HostTensor* bp = get(pre + ".proj_out.bias", {C});
    const HostTensor* bv = get(pre + ".v.bias", {C});
    if (!wp || !bp || !bv) return false;
    a.proj.N = C; a.proj.K = C;
    a.proj.w = upload16(C, C, [&](size_t r, size_t c) { return wp->at(r * C + c); });
    a.proj.b = upload32(C, [&](size_t i) {
        double acc = bp->at(i);
        for (int k = 0; k < C; ++k) acc += (double)wp->at(i * C + k) * (double)bv->at(k);
        return (float)acc;
    });
    if (!a.proj.w || !a.proj.b) return false;
    if (C == 512) {       // the flash kernel's operands: one [3C][C] projection, q | k | v rows as attention.hip's fused q|k|v (V row-major: no V^T GEMM)
        const HostTensor *wq = get(pre + ".q.weight", {C, C, 1, 1}), *wk = get(pre + ".k.weight", {C, C, 1, 1}), *wv = get(pre + ".v.weight", {C, C, 1, 1});
        const HostTensor *bq = get(pre + ".q.bias", {C}), *bk = get(pre + ".k.bias", {C});
        if (!wq || !wk || !wv || !bq || !bk) return false;
        a.qkv.N = 3 * C; a.qkv.K = C;
        a.qkv.w = upload16((size_t)3 * C, C, [&](size_t r, size_t c) { const HostTensor* s = r < (size_t)C ? wq : (r < (size_t)2 * C ? wk : wv); return s->at((r % C) * C + c); });
        a.qkv.b = upload32((size_t)3 * C, [&](size_t i) { return i < (size_t)C ? bq->at(i) : (i < (size_t)2 * C ? bk->at(i - C) : 0.f); });
        if (!a.qkv.w || !a.qkv.b) return false;
    }
    return true;
}

int Engine::finalize_vae() {
    if (finalized) return LDX_OK;
    const ldx_vae_config& v = vcfg;
    auto bad = [&](const char* m) { set_error(std::string("unsupported VAE config: ") + m); return LDX_EINVAL; };
    if (v.ch <= 0 || v.ch % 64) return bad("ch must be a multiple of 64");
    if (v.num_levels < 1 || v.num_levels > 8 || v.z_channels < 1 || v.z_channels > 16 || v.out_ch < 1 || v.out_ch > 4) return bad("levels/channels");
    HIP_OK(hipSetDevice(device));
    bool ok = true;
    int block_in = v.ch * v.ch_mult[v.num_levels - 1];
    ok = ok && mk_conv3("decoder.conv_in", block_in, v.z_channels, 64, conv_in);
    ok = ok && mk_vae_res("decoder.mid.block_1", block_in, block_in, vae_mid1);
    ok = ok && mk_vae_attn("decoder.mid.attn_1", block_in, vae_attn);
    ok = ok && mk_vae_res("decoder.mid.block_2", block_in, block_in, vae_mid2);
    vae_up.assign(v.num_levels, {}); vae_upconv.assign(v.num_levels, LinearW{}); vae_has_up.assign(v.num_levels, false);
    for (int lv = v.num_levels - 1; ok && lv >= 0; --lv) {
        const int block_out = v.ch * v.ch_mult[lv];
        for (int i = 0; ok && i <= v.num_res_blocks; ++i) {
            ResW r;
            ok = mk_vae_res("decoder.up." + std::to_string(lv) + ".block." + std::to_string(i), block_in, block_out, r);
            vae_up[lv].push_back(r);
            block_in = block_out;
        }
        if (ok && lv != 0) {
            vae_has_up[lv] = true;
            ok = mk_conv3("decoder.up." + std::to_string(lv) + ".upsample.conv", block_in, block_in, block_in, vae_upconv[lv]);
        }
    }
    ok = ok && mk_norm("decoder.norm_out", block_in, vae_norm_out) && mk_conv3("decoder.conv_out", v.out_ch, block_in, block_in, conv_out);
    if (ok && v.use_post_quant) {
        const HostTensor* w = get("post_quant_conv.weight", {v.z_channels, v.z_channels, 1, 1});
        const HostTensor* b = get("post_quant_conv.bias", {v.z_channels});
        ok = w && b;
        if (ok) {
            const int zc = v.z_channels;
            vae_pq = upload32((size_t)zc * zc + zc, [&](size_t i) { return i < (size_t)zc * zc ? w->at(i) : b->at(i - (size_t)zc * zc); });
            ok = vae_pq != nullptr;
        }
    }
    // ---- optional encoder (Encoder.__init__, VariationalAE.py:257-377) ----
    if (ok && host.count("encoder.conv_in.weight")) {
        vae_has_enc = true;
        const int in_px = 3;
        ok = mk_conv3("encoder.conv_in", v.ch, in_px, 64, enc_conv_in);
        enc_down.assign(v.num_levels, {}); enc_downconv.assign(v.num_levels, LinearW{});
        int bin = v.ch;
        for (int lv = 0; ok && lv < v.num_levels; ++lv) {
            const int bout = v.ch * v.ch_mult[lv];
            for (int i = 0; ok && i < v.num_res_blocks; ++i) {
                ResW r;
                ok = mk_vae_res("encoder.down." + std::to_string(lv) + ".block." + std::to_string(i), bin, bout, r);
                enc_down[lv].push_back(r);
                bin = bout;
            }
            if (ok && lv != v.num_levels - 1) ok = mk_conv3("encoder.down." + std::to_string(lv) + ".downsample.conv", bin, bin, bin, enc_downconv[lv]);
        }
        ok = ok && mk_vae_res("encoder.mid.block_1", bin, bin, enc_mid1) && mk_vae_attn("encoder.mid.attn_1", bin, enc_attn) &&
             mk_vae_res("encoder.mid.block_2", bin, bin, enc_mid2) && mk_norm("encoder.norm_out", bin, enc_norm_out) &&
             mk_conv3("encoder.conv_out", 2 * v.z_channels, bin, bin, enc_conv_out);
        if (ok && v.use_post_quant) {
            const int zc2 = 2 * v.z_channels;
            const HostTensor* w = get("quant_conv.weight", {zc2, zc2, 1, 1});
            const HostTensor* b = get("quant_conv.bias", {zc2});
            ok = w && b;
            if (ok) { enc_qc = upload32((size_t)zc2 * zc2 + zc2, [&](size_t i) { return i < (size_t)zc2 * zc2 ? w->at(i) : b->at(i - (size_t)zc2 * zc2); }); ok = enc_qc != nullptr; }
        }
    }
    if (!ok) {
        if (!missing.empty()) { set_error("missing or mis-shaped weight: " + missing); return LDX_EMISSING; }
        set_error(std::string("weight upload failed: ") + hipGetErrorString(hipGetLastError()));
        return LDX_EHIP;
    }
    host.clear();
    finalized = true;
    return LDX_OK;
}

// AttnBlock: x + proj(softmax(q k^T / sqrt(C)) v), one head of width C, as three GEMMs + a row softmax
void Engine::emit_vae_attn(const VaeAttnW& a, Act X, Act OUT, int B, int H, int W) {
    const int N = H * W, M = B * N, C = a.q.N;
    Act hn = new_act(M, C);
    op_gn("vae.attn.norm", X, hn, B, N, a.norm, 1e-6f, false);
    {
        // C = 512 (every SD VAE): flash attention, one head of D = 512 (attn512.hip) — q | k | v in one projection, the scores stay in the CUs.
        // Before round 5 (and still for other widths / LDX_ATTN512=0): q k^T GEMM -> row softmax -> p v GEMM through HBM, below.
        AttnArgs probe{}; probe.D = C; probe.B = B; probe.H = 1; probe.Nq = N; probe.Mk = N; probe.ldq = probe.ldk = probe.ldv = 3 * C; probe.ldo = C;
        if (a.qkv.w && attn512_ok(probe)) {
            Act qkv = new_act(M, 3 * C);
            op_gemm("vae.attn.qkv", hn, a.qkv, qkv, Act{});
            release(hn);
            Act o = new_act(M, C);
            const char* base = (const char*)ptr(qkv);
            op_attn("vae.attn.flash", base, 3 * C, base + (size_t)C * 2, 3 * C, base + (size_t)2 * C * 2, 3 * C, o, B, 1, N, N, C);
            release(qkv);
            op_gemm("vae.attn.proj", o, a.proj, OUT, X);
            release(o);
            return;
        }
    }
    Act q = new_act(M, C), k = new_act(M, C);
    op_gemm("vae.attn.q", hn, a.q, q, Act{});
    op_gemm("vae.attn.k", hn, a.k, k, Act{});
    Act o = new_act(M, C);
    for (int b = 0; b < B; ++b) {
        auto rows = [&](const Act& t, int r0, int nr) { Act v = t; v.owned = false; v.off = t.off + (size_t)r0 * t.ld * 2; v.rows = nr; return v; };
        // V^T[C][N] = Wv[C][C] . hn_b[N][C]^T   (weights as the A operand, activations as "W")
        Act vt = new_act(C, N);
        { LinearW hw; hw.w = ptr(rows(hn, b * N, N)); hw.b = nullptr; hw.N = N; hw.K = C;
          Act wv; wv.valid = true; wv.rows = C; wv.C = C; wv.ld = C; wv.off = 0; wv.col = 0;
          Op oo{}; oo.kind = OP_GEMM; oo.name = "vae.attn.vT";
          GemmArgs& g = oo.g; g.A = a.v.w; g.lda = C; g.W = hw.w; g.M = C; g.N = N; g.K = C; g.mode = 0; g.rows_per_batch = 1;
          g.C = ptr(vt); g.ldc = N; g.splitk = 1;
          oo.flops = 2.0 * C * (double)N * C; snprintf(oo.klabel, sizeof(oo.klabel), "gemm_kernel<%s,0>", dt == DT_BF16 ? "bf16" : "f16");
          ops.push_back(oo); flops += oo.flops; (void)wv; }
        // Query rows in chunks of Rc: S_c[Rc][N] = q_c k_b^T, row softmax, O_c = P_c V.  The score matrix is never materialised as a whole
        // (N x N at 2048^2 is 8 GiB and was the arena's peak; at 4096^2 it would be 128 GiB): one chunk is <= 2 GiB, so a 1024^2 decode still runs
        // the three launches it always ran and a 2048^2 decode 4 x 3 (measured at 2048^2: one chunk 69.3 ms, 512 MiB chunks 71.9, 128 MiB 72.1; the
        // arena peak then sits at the 2048^2 x 128-channel conv level, 6.5 GiB instead of 8.45).  LDX_VAE_ATTN_CHUNK_MIB overrides (0 = one chunk).
        static const long chunk_mib = getenv("LDX_VAE_ATTN_CHUNK_MIB") ? atol(getenv("LDX_VAE_ATTN_CHUNK_MIB")) : 2048;
        int Rc = N;
        if (chunk_mib > 0) {
            long r = (chunk_mib << 20) / ((long)N * 2);
            r = (r / 256) * 256;
            if (r < 256) r = 256;
            if (r < N) Rc = (int)r;
        }
        Act S = new_act(Rc, N);
        for (int r0 = 0; r0 < N; r0 += Rc) {
            const int nr = std::min(Rc, N - r0);
            Act Sc = rows(S, 0, nr);
            { LinearW kw; kw.w = ptr(rows(k, b * N, N)); kw.b = nullptr; kw.N = N; kw.K = C;
              op_gemm("vae.attn.qk", rows(q, b * N + r0, nr), kw, Sc, Act{}); }
            { Op oo{}; oo.kind = OP_SOFTMAX; oo.name = "vae.attn.softmax"; oo.p1 = ptr(Sc); oo.i0 = nr; oo.i1 = N; oo.i2 = N; oo.f0 = 1.0f / std::sqrt((float)C);
              oo.bytes = 2.0 * 2.0 * nr * (double)N; snprintf(oo.klabel, sizeof(oo.klabel), "softmax_rows"); ops.push_back(oo); }
            // O_c[nr][C] = P_c[nr][N] . V[N][C]  with W = V^T[C][N]
            { LinearW vw; vw.w = ptr(vt); vw.b = nullptr; vw.N = C; vw.K = N;
              op_gemm("vae.attn.pv", Sc, vw, rows(o, b * N + r0, nr), Act{}); }
        }
        release(S); release(vt);
    }
    release(q); release(k); release(hn);
    op_gemm("vae.attn.proj", o, a.proj, OUT, X);
    release(o);
}

int Engine::plan_vae(int B, int h, int w) {
    const ldx_vae_config& v = vcfg;
    for (int pass = 0; pass < 2; ++pass) {
        ops.clear(); flops = 0; free_list.clear(); live.clear(); arena_top = 0; arena_peak = 0;
        if (pass == 1) {
            if (arena && arena_cap < arena_peak_dry) { HIP_OK(hipFree(arena)); arena = nullptr; }
            if (!arena) { HIP_OK(hipMalloc(&arena, arena_peak_dry)); arena_cap = arena_peak_dry; }
        }
        void* saved = arena;
        if (pass == 0) arena = nullptr;
        gn_ws_off = a_alloc(gn_ws_bytes(B, (long)h * w * 64));
        int H = h, W = w;
        Act x0 = new_act(B * H * W, 64);
        { Op o{}; o.kind = OP_VAEPREP; o.name = "vae.prep"; o.p1 = ptr(x0); o.i0 = B; o.i1 = v.z_channels; o.i2 = H * W; o.i3 = 64; ops.push_back(o); }
        int C = v.ch * v.ch_mult[v.num_levels - 1];
        Act hcur = new_act(B * H * W, C);
        op_conv("vae.conv_in", x0, B, H, W, 64, conv_in, 1, H, W, hcur, Act{});
        flops -= 2.0 * B * H * W * (double)C * 9.0 * (64 - v.z_channels);
        release(x0);
        auto res = [&](const ResW& r) { Act o = new_act(B * H * W, r.Cout); emit_res(r, hcur, o, B, H, W); release(hcur); hcur = o; };
        res(vae_mid1);
        { Act o = new_act(B * H * W, C); emit_vae_attn(vae_attn, hcur, o, B, H, W); release(hcur); hcur = o; }
        res(vae_mid2);
        for (int lv = v.num_levels - 1; lv >= 0; --lv) {
            for (auto& r : vae_up[lv]) res(r);
            if (vae_has_up[lv]) {
                const int Cc = vae_up[lv].back().Cout;
                Act o = new_act(B * 2 * H * 2 * W, Cc);
                op_conv("vae.up", hcur, B, H, W, Cc, vae_upconv[lv], 1, 2 * H, 2 * W, o, Act{});
                release(hcur); hcur = o; H *= 2; W *= 2;
            }
        }
        const int Cl = vae_up[0].back().Cout;
        Act t = new_act(B * H * W, Cl);
        op_gn("vae.norm_out", hcur, t, B, H * W, vae_norm_out, 1e-6f, true);
        release(hcur);
        const size_t o_pix = a_alloc((size_t)B * H * W * v.out_ch * 4);
        float* pix = (float*)((uintptr_t)arena + o_pix);
        op_conv("vae.conv_out", t, B, H, W, Cl, conv_out, 1, H, W, Act{}, Act{}, nullptr, 0, pix, v.out_ch);
        release(t);
        { Op o{}; o.kind = OP_CLAMP; o.name = "vae.clamp"; o.p0 = pix; o.i0 = B * H * W * v.out_ch; ops.push_back(o); }
        fuse_gn_stats();
        if (pass == 0) { arena_peak_dry = arena_peak; arena = saved; }
    }
    pB2 = B; ph = h; pw = w; pM = 0; vae_plan_mode = 1;
    return LDX_OK;
}

int Engine::run_vae(const float* z, int B, int h, int w, float* out, hipStream_t st) {
    if (!finalized || kind != KIND_VAE) { set_error("ldx_vae_decode: not a finalized VAE engine"); return LDX_ESTATE; }
    if (!z || !out || B <= 0 || h <= 0 || w <= 0) { set_error("ldx_vae_decode: bad argument"); return LDX_EINVAL; }
    if ((h * w) % 8) { set_error("ldx_vae_decode: h*w must be a multiple of 8 (attention row length)"); return LDX_EINVAL; }
    HIP_OK(hipSetDevice(device));
    if (B != pB2 || h != ph || w != pw || vae_plan_mode != 1) {
        HIP_OK(hipStreamSynchronize(st));
        int rc = plan_vae(B, h, w);
        if (rc) return rc;
    }
    b_x = z; b_out = out; prof_graph = false;
    int rc = exec_ops(st);
    if (rc) return rc;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error(std::string("kernel launch: ") + hipGetErrorString(e)); return LDX_EHIP; }
    return LDX_OK;
}

// Encoder.forward (VariationalAE.py:378-413) + quant_conv: pixels -> moments
int Engine::plan_vae_encode(int B, int Hpx, int Wpx) {
    const ldx_vae_config& v = vcfg;
    for (int pass = 0; pass < 2; ++pass) {
        ops.clear(); flops = 0; free_list.clear(); live.clear(); arena_top = 0; arena_peak = 0;
        if (pass == 1) {
            if (arena && arena_cap < arena_peak_dry) { HIP_OK(hipFree(arena)); arena = nullptr; }
            if (!arena) { HIP_OK(hipMalloc(&arena, arena_peak_dry)); arena_cap = arena_peak_dry; }
        }
        void* saved = arena;
        if (pass == 0) arena = nullptr;
        gn_ws_off = a_alloc(gn_ws_bytes(B, (long)Hpx * Wpx));
        int H = Hpx, W = Wpx;
        Act x0 = new_act(B * H * W, 64);
        { Op o{}; o.kind = OP_PIXPREP; o.name = "vae.enc.prep"; o.p1 = ptr(x0); o.i0 = B; o.i1 = 3; o.i2 = H * W; o.i3 = 64; o.f0 = 2.0f; o.f1 = -1.0f; ops.push_back(o); }
        Act hcur = new_act(B * H * W, v.ch);
        op_conv("vae.enc.conv_in", x0, B, H, W, 64, enc_conv_in, 1, H, W, hcur, Act{});
        flops -= 2.0 * B * H * W * (double)v.ch * 9.0 * (64 - 3);
        release(x0);
        auto res = [&](const ResW& r) { Act o = new_act(B * H * W, r.Cout); emit_res(r, hcur, o, B, H, W); release(hcur); hcur = o; };
        for (int lv = 0; lv < v.num_levels; ++lv) {
            for (auto& r : enc_down[lv]) res(r);
            if (lv != v.num_levels - 1) {
                // Downsample: F.pad(x, (0,1,0,1)) then 3x3 stride-2 conv without padding (VariationalAE.py:224-254)
                const int Cc = enc_down[lv].back().Cout, Ho = H / 2, Wo = W / 2;
                Act o = new_act(B * Ho * Wo, Cc);
                op_conv("vae.enc.down", hcur, B, H, W, Cc, enc_downconv[lv], 2, Ho, Wo, o, Act{});
                ops.back().g.pad0 = 1;
                release(hcur); hcur = o; H = Ho; W = Wo;
            }
        }
        res(enc_mid1);
        { Act o = new_act(B * H * W, enc_mid1.Cout); emit_vae_attn(enc_attn, hcur, o, B, H, W); release(hcur); hcur = o; }
        res(enc_mid2);
        const int Cl = enc_mid2.Cout, zc2 = 2 * v.z_channels;
        Act t = new_act(B * H * W, Cl);
        op_gn("vae.enc.norm_out", hcur, t, B, H * W, enc_norm_out, 1e-6f, true);
        release(hcur);
        const size_t o_m = a_alloc((size_t)B * H * W * zc2 * 4);
        float* mom = (float*)((uintptr_t)arena + o_m);
        op_conv("vae.enc.conv_out", t, B, H, W, Cl, enc_conv_out, 1, H, W, Act{}, Act{}, nullptr, 0, mom, zc2);
        release(t);
        { Op o{}; o.kind = OP_MOMENTS; o.name = "vae.enc.quant_conv"; o.p0 = mom; o.i0 = B; o.i1 = zc2; o.i2 = H * W; ops.push_back(o); }
        fuse_gn_stats();
        if (pass == 0) { arena_peak_dry = arena_peak; arena = saved; }
    }
    pB2 = B; ph = Hpx; pw = Wpx; pM = 0; vae_plan_mode = 2;
    return LDX_OK;
}

int Engine::run_vae_encode(const float* px, int B, int H, int W, float* moments, hipStream_t st) {
    if (!finalized || kind != KIND_VAE) { set_error("ldx_vae_encode: not a finalized VAE engine"); return LDX_ESTATE; }
    if (!vae_has_enc) { set_error("ldx_vae_encode: encoder.* weights were not loaded"); return LDX_EMISSING; }
    const int f = 1 << (vcfg.num_levels - 1);
    // sizes that are not multiples of f are legal in the reference (vae_encode_crop_pixels is a no-op, VariationalAE.py:
    // 677-688): every Downsample yields floor(H/2)
    if (!px || !moments || B <= 0 || H < f || W < f || ((H / f) * (W / f)) % 8) {
        set_error("ldx_vae_encode: bad argument (H, W >= downscale factor; latent h*w multiple of 8)"); return LDX_EINVAL; }
    HIP_OK(hipSetDevice(device));
    if (B != pB2 || H != ph || W != pw || vae_plan_mode != 2) {
        HIP_OK(hipStreamSynchronize(st));
        int rc = plan_vae_encode(B, H, W);
        if (rc) return rc;
    }
    b_x = px; b_out = moments; prof_graph = false;
    int rc = exec_ops(st);
    if (rc) return rc;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error(std::string("kernel launch: ") + hipGetErrorString(e)); return LDX_EHIP; }
    return LDX_OK;
}

// =============================================================================================
// CLIP
int Engine::finalize_clip() {
    if (finalized) return LDX_OK;
    const ldx_clip_config& c = ccfg;
    auto bad = [&](const char* m) { set_error(std::string("unsupported CLIP config: ") + m); return LDX_EINVAL; };
    if (c.hidden_size % 64 || c.intermediate_size % 64) return bad("hidden/intermediate size must be multiples of 64");
    if (c.hidden_size % c.num_heads || (c.hidden_size / c.num_heads) % 8 || c.hidden_size / c.num_heads > 160) return bad("head dim");
    HIP_OK(hipSetDevice(device));
    const int E = c.hidden_size;
    bool ok = true;
    const HostTensor* tok = get("embeddings.token_embedding.weight", {c.vocab_size, E});
    const HostTensor* pos = get("embeddings.position_embedding.weight", {c.max_positions, E});
    ok = tok && pos;
    if (ok) {
        clip_tok = upload32((size_t)c.vocab_size * E, [&](size_t i) { return tok->at(i); });
        clip_pos = upload32((size_t)c.max_positions * E, [&](size_t i) { return pos->at(i); });
        ok = clip_tok && clip_pos;
    }
    clip_layers.resize(c.num_layers);
    for (int l = 0; ok && l < c.num_layers; ++l) {
        ClipLayerW& L = clip_layers[l];
        const std::string p = "encoder.layers." + std::to_string(l);
        ok = mk_norm(p + ".layer_norm1", E, L.ln1) && mk_norm(p + ".layer_norm2", E, L.ln2);
        const HostTensor *qw = get(p + ".self_attn.q_proj.weight", {E, E}), *kw = get(p + ".self_attn.k_proj.weight", {E, E}),
                         *vw = get(p + ".self_attn.v_proj.weight", {E, E}), *qb = get(p + ".self_attn.q_proj.bias", {E}),
                         *kb = get(p + ".self_attn.k_proj.bias", {E}), *vb = get(p + ".self_attn.v_proj.bias", {E});
        ok = ok && qw && kw && vw && qb && kb && vb;
        if (ok) {
            L.qkv.N = 3 * E; L.qkv.K = E;
            L.qkv.w = upload16((size_t)3 * E, E, [&](size_t r, size_t cc) { const HostTensor* s = r < (size_t)E ? qw : (r < (size_t)2 * E ? kw : vw); return s->at((r % E) * E + cc); });
            L.qkv.b = upload32((size_t)3 * E, [&](size_t i) { const HostTensor* s = i < (size_t)E ? qb : (i < (size_t)2 * E ? kb : vb); return s->at(i % E); });
            ok = L.qkv.w && L.qkv.b;
        }
        ok = ok && mk_linear(p + ".self_attn.out_proj", E, E, true, L.out) && mk_linear(p + ".mlp.fc1", c.intermediate_size, E, true, L.fc1) &&
             mk_linear(p + ".mlp.fc2", E, c.intermediate_size, true, L.fc2);
    }
    ok = ok && mk_norm("final_layer_norm", E, clip_final_ln);
    if (ok && host.count("text_projection.weight")) {          // optional: the pooled output's projection (bias-free Linear)
        const HostTensor* tp = get("text_projection.weight", {E, E});
        ok = tp != nullptr;
        if (ok) { clip_proj = upload32((size_t)E * E, [&](size_t i) { return tp->at(i); }); ok = clip_proj != nullptr; }
    }
    if (!ok) {
        if (!missing.empty()) { set_error("missing or mis-shaped weight: " + missing); return LDX_EMISSING; }
        set_error(std::string("weight upload failed: ") + hipGetErrorString(hipGetLastError()));
        return LDX_EHIP;
    }
    host.clear();
    finalized = true;
    return LDX_OK;
}

int Engine::plan_clip(int B, int T, int inter) {
    const ldx_clip_config& c = ccfg;
    const int E = c.hidden_size, M = B * T, heads = c.num_heads, D = E / heads;
    for (int pass = 0; pass < 2; ++pass) {
        ops.clear(); flops = 0; free_list.clear(); live.clear(); arena_top = 0; arena_peak = 0;
        if (pass == 1) {
            if (arena && arena_cap < arena_peak_dry) { HIP_OK(hipFree(arena)); arena = nullptr; }
            if (!arena) { HIP_OK(hipMalloc(&arena, arena_peak_dry)); arena_cap = arena_peak_dry; }
        }
        void* saved = arena;
        if (pass == 0) arena = nullptr;
        Act x = new_act(M, E);
        { Op o{}; o.kind = OP_EMBED; o.name = "clip.embed"; o.p1 = ptr(x); o.i0 = B; o.i1 = T; o.i2 = E; o.i3 = c.vocab_size; ops.push_back(o); }
        Act n = new_act(M, E), qkv = new_act(M, 3 * E), a = new_act(M, E), f = new_act(M, c.intermediate_size);
        Act xi = new_act(M, E);
        for (int l = 0; l < c.num_layers; ++l) {
            const ClipLayerW& L = clip_layers[l];
            op_ln("clip.ln1", x, n, L.ln1);
            op_gemm("clip.qkv", n, L.qkv, qkv, Act{});
            const char* base = (const char*)ptr(qkv);
            op_attn("clip.attn", base, 3 * E, base + (size_t)E * 2, 3 * E, base + (size_t)2 * E * 2, 3 * E, a, B, heads, T, T, D);
            ops.back().at.causal = 1;
            op_gemm("clip.out", a, L.out, x, x);                 // x += self_attn(ln1(x))
            op_ln("clip.ln2", x, n, L.ln2);
            op_gemm("clip.fc1", n, L.fc1, f, Act{});
            ops.back().g.act = 1;                                // quick-GELU
            op_gemm("clip.fc2", f, L.fc2, x, x);                 // x += mlp(ln2(x))
            if (l == inter) {                                    // intermediate = x.clone(); final LN applied to it
                op_ln("clip.final_ln.inter", x, xi, clip_final_ln);
                Op o{}; o.kind = OP_CVT_OUT; o.name = "clip.out_inter"; o.p0 = ptr(xi); o.i0 = M * E; o.i3 = 1; ops.push_back(o);
            }
        }
        op_ln("clip.final_ln", x, n, clip_final_ln);
        { Op o{}; o.kind = OP_CVT_OUT; o.name = "clip.out_last"; o.p0 = ptr(n); o.i0 = M * E; o.i3 = 0; ops.push_back(o); }
        if (pass == 0) { arena_peak_dry = arena_peak; arena = saved; }
    }
    pB2 = B; ph = T; pw = 0; pM = 0; clip_inter_planned = inter;
    return LDX_OK;
}

// Textual-inversion vectors for the next ldx_clip_encode calls: the reference extends the token table by one row per vector
// and gives those rows the ids vocab_size, vocab_size + 1, ... (SDClipModel.set_up_textual_embeddings, SD15/SDClip.py:213-267).
// Host rows [n][hidden] fp32; n = 0 removes them.  Synchronous (cudaMemcpy); the buffer grows, never shrinks.
int Engine::set_clip_extra(const float* rows_host, int n) {
    if (!finalized || kind != KIND_CLIP) { set_error("ldx_clip_set_extra_embeddings: not a finalized CLIP engine"); return LDX_ESTATE; }
    if (n < 0 || (n > 0 && !rows_host)) { set_error("ldx_clip_set_extra_embeddings: bad argument"); return LDX_EINVAL; }
    HIP_OK(hipSetDevice(device));
    HIP_OK(hipDeviceSynchronize());                    // a previous encode may still read the old rows
    const size_t E = (size_t)ccfg.hidden_size;
    if (n > clip_extra_cap) {
        float* p = nullptr;
        HIP_OK(hipMalloc((void**)&p, (size_t)n * E * sizeof(float)));
        dev_allocs.push_back(p);                       // the old (smaller) buffer stays owned by the engine until it is destroyed
        clip_extra = p; clip_extra_cap = n;
    }
    if (n > 0) HIP_OK(hipMemcpy(clip_extra, rows_host, (size_t)n * E * sizeof(float), hipMemcpyHostToDevice));
    clip_extra_n = n;
    return LDX_OK;
}

int Engine::clip_pooled(const float* last, const int* ids, int B, int T, int eos_id, float* out, hipStream_t st) {
    if (kind != KIND_CLIP || !finalized) { set_error("ldx_clip_pooled: not a finalized CLIP engine"); return LDX_ESTATE; }
    if (!last || !ids || !out || B <= 0 || T <= 0) { set_error("ldx_clip_pooled: bad argument"); return LDX_EINVAL; }
    HIP_OK(hipSetDevice(device));
    launch_clip_pooled(last, ids, B, T, ccfg.hidden_size, eos_id, clip_proj, out, st);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error(std::string("kernel launch: ") + hipGetErrorString(e)); return LDX_EHIP; }
    return LDX_OK;
}

int Engine::run_clip(const int* ids, int B, int T, int inter_layer, float* out_last, float* out_inter, hipStream_t st) {
    if (!finalized || kind != KIND_CLIP) { set_error("ldx_clip_encode: not a finalized CLIP engine"); return LDX_ESTATE; }
    if (!ids || !out_last || B <= 0 || T <= 0 || T > ccfg.max_positions) { set_error("ldx_clip_encode: bad argument"); return LDX_EINVAL; }
    HIP_OK(hipSetDevice(device));
    int inter = -1;
    if (out_inter) {
        inter = inter_layer < 0 ? ccfg.num_layers + inter_layer : inter_layer;
        if (inter < 0 || inter >= ccfg.num_layers) { set_error("ldx_clip_encode: inter_layer out of range"); return LDX_EINVAL; }
    }
    if (B != pB2 || T != ph || inter != clip_inter_planned) {
        HIP_OK(hipStreamSynchronize(st));
        int rc = plan_clip(B, T, inter);
        if (rc) return rc;
    }
    b_ids = ids; b_out = out_last; b_out2 = out_inter; prof_graph = false;
    int rc = exec_ops(st);
    if (rc) return rc;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error(std::string("kernel launch: ") + hipGetErrorString(e)); return LDX_EHIP; }
    return LDX_OK;
}

}  // namespace ldx
