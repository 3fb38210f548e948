// ESRGAN RRDBNet (x2^n super-resolution) as a static launch plan over the implicit-GEMM conv kernel.
// Reference: src/UltimateSDUpscale/RDRB.py — RRDBNet :216-471, RRDB :10-77, ResidualDenseBlock_5C :80-205;
// blocks from USDU_util.py (conv_block :36-98, upconv_block :101-128, ShortcutBlock :131-138).  Keys are the module's own
// ("old arch") names: model.0, model.1.sub.<i>.RDB<k>.conv<j>.0, model.1.sub.<nb>, model.3 / model.6 (upconv), model.8, model.10.
//
// A dense block's torch.cat((x, x1, ..)) is a column range of one [M][nf + 4*gc] buffer: conv_j reads the first nf + (j-1)*gc
// columns (rounded up to a multiple of 64 with zero weights) and writes its gc outputs behind them.  conv5's epilogue does
// x5 * 0.2 + x, and for RDB3 also the RRDB's  out * 0.2 + x  (second residual), so no elementwise pass exists.
// Nearest x2 upsampling is fused into the following conv's gather.
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

Engine::Engine(const ldx_esrgan_config& c, int dev) : cfg{}, device(dev) {
    kind = KIND_ESRGAN; ecfg = c;
    dt = (c.compute_dtype == LDX_F16) ? DT_F16 : DT_BF16;
}

int Engine::finalize_esrgan() {
    if (finalized) return LDX_OK;
    const ldx_esrgan_config& c = ecfg;
    auto bad = [&](const char* m) { set_error(std::string("unsupported ESRGAN config: ") + m); return LDX_EINVAL; };
    if (c.nf != 64 || c.gc != 32) return bad("nf must be 64 and gc 32 (the only RRDBNet layout the reference builds, RDRB.py:283-296)");
    if (c.in_nc <= 0 || c.in_nc > 64 || c.out_nc <= 0 || c.out_nc > 64 || c.num_blocks <= 0 || c.num_upscale < 0 || c.num_upscale > 4) return bad("channels / blocks / scale");
    HIP_OK(hipSetDevice(device));
    const int nf = c.nf, gc = c.gc;
    bool ok = mk_conv3("model.0", nf, c.in_nc, 64, es_first);
    es_rdb.resize((size_t)c.num_blocks * 3);
    for (int i = 0; ok && i < c.num_blocks; ++i)
        for (int k = 0; ok && k < 3; ++k) {
            RdbW& r = es_rdb[(size_t)i * 3 + k];
            const std::string p = "model.1.sub." + std::to_string(i) + ".RDB" + std::to_string(k + 1) + ".conv";
            for (int j = 0; ok && j < 5; ++j) {
                const int cin = nf + j * gc, cpad = (cin + 63) / 64 * 64;
                ok = mk_conv3(p + std::to_string(j + 1) + ".0", j < 4 ? gc : nf, cin, cpad, r.c[j]);
            }
        }
    ok = ok && mk_conv3("model.1.sub." + std::to_string(c.num_blocks), nf, nf, nf, es_trunk);
    es_up.resize(c.num_upscale);
    for (int u = 0; ok && u < c.num_upscale; ++u) ok = mk_conv3("model." + std::to_string(3 * (u + 1)), nf, nf, nf, es_up[u]);
    ok = ok && mk_conv3("model." + std::to_string(3 * c.num_upscale + 2), nf, nf, nf, es_hr) &&
         mk_conv3("model." + std::to_string(3 * c.num_upscale + 4), c.out_nc, nf, nf, es_last);
    if (!ok) {
        if (!missing.empty()) { set_error("missing or mis-shaped weight: " + missing); return LDX_EMISSING; }
        set_error(std::string("weight upload failed: ") + hipGetErrorString(hipGetLastError()));
        return LDX_EHIP;
    }
    host.clear();
    finalized = true;
    return LDX_OK;
}

int Engine::plan_esrgan(int B, int H, int W) {
    const ldx_esrgan_config& c = ecfg;
    const int nf = c.nf, gc = c.gc, CW = nf + 4 * gc, M = B * H * W;
    for (int pass = 0; pass < 2; ++pass) {
        ops.clear(); flops = 0; free_list.clear(); live.clear(); arena_top = 0; arena_peak = 0;
        if (pass == 1) {
            if (arena && arena_cap < arena_peak_dry) { HIP_OK(hipFree(arena)); arena = nullptr; }
            if (!arena) { HIP_OK(hipMalloc(&arena, arena_peak_dry)); arena_cap = arena_peak_dry; }
            // dense-block convs read a few not-yet-written columns against zero weights: those must hold finite values, and a
            // new plan lays the buffers over whatever the previous shape left there (fp32 pixels read as 16-bit can be NaN)
            HIP_OK(hipMemset(arena, 0, arena_cap));
        }
        void* saved = arena;
        if (pass == 0) arena = nullptr;
        auto conv = [&](const char* name, Act X, int Cin, const LinearW& w, int Hin, int Win, int Hout, int Wout, Act Y, Act R, int act) {
            op_conv(name, X, B, Hin, Win, Cin, w, 1, Hout, Wout, Y, R);
            GemmArgs& g = ops.back().g; g.act = act;
            if (g.splitk > 1) g.splitk = 1;
        };
        Act x0 = new_act(M, 64);
        { Op o{}; o.kind = OP_PIXPREP; o.name = "esrgan.prep"; o.p1 = ptr(x0); o.i0 = B; o.i1 = c.in_nc; o.i2 = H * W; o.i3 = 64; o.f0 = 1.0f; o.f1 = 0.0f; ops.push_back(o); }
        Act fea = new_act(M, nf);
        Act cat[3] = {new_act(M, CW), new_act(M, CW), new_act(M, CW)};
        conv("esrgan.conv_first", x0, 64, es_first, H, W, H, W, fea, Act{}, 0);
        conv("esrgan.conv_first", x0, 64, es_first, H, W, H, W, view(cat[0], 0, nf), Act{}, 0);      // second copy feeds the first dense block
        release(x0);
        int in = 0, t1 = 1, t2 = 2;
        auto rdb = [&](const RdbW& r, int src, int dst, const Act* rrdb_in) {
            for (int j = 0; j < 4; ++j) {
                const int cpad = (nf + j * gc + 63) / 64 * 64;
                conv("esrgan.rdb.conv", view(cat[src], 0, cpad), cpad, r.c[j], H, W, H, W, view(cat[src], nf + j * gc, gc), Act{}, 3);
            }
            conv("esrgan.rdb.conv5", view(cat[src], 0, CW), CW, r.c[4], H, W, H, W, view(cat[dst], 0, nf), view(cat[src], 0, nf), 0);
            GemmArgs& g = ops.back().g; g.oscale = 0.2f;                                      // x5 * 0.2 + x
            if (rrdb_in) { g.R2 = ptr(*rrdb_in); g.ldr2 = rrdb_in->ld; g.oscale2 = 0.2f; }   // (..) * 0.2 + rrdb input
        };
        for (int i = 0; i < c.num_blocks; ++i) {
            const Act xin = view(cat[in], 0, nf);
            rdb(es_rdb[(size_t)i * 3 + 0], in, t1, nullptr);
            rdb(es_rdb[(size_t)i * 3 + 1], t1, t2, nullptr);
            rdb(es_rdb[(size_t)i * 3 + 2], t2, t1, &xin);
            const int nin = t1; t1 = t2; t2 = in; in = nin;
        }
        Act trunk = new_act(M, nf);
        conv("esrgan.trunk_conv", view(cat[in], 0, nf), nf, es_trunk, H, W, H, W, trunk, fea, 0);    // ShortcutBlock: fea + trunk(fea)
        release(cat[0]); release(cat[1]); release(cat[2]); release(fea);
        Act cur = trunk;
        int h = H, w = W;
        for (int u = 0; u < c.num_upscale; ++u) {
            Act nx = new_act(B * 4 * h * w, nf);
            conv("esrgan.upconv", cur, nf, es_up[u], h, w, 2 * h, 2 * w, nx, Act{}, 3);
            release(cur); cur = nx; h *= 2; w *= 2;
        }
        Act hr = new_act(B * h * w, nf);
        conv("esrgan.hr_conv", cur, nf, es_hr, h, w, h, w, hr, Act{}, 3);
        release(cur);
        const size_t o_pix = a_alloc((size_t)B * h * w * c.out_nc * 4);
        float* pix = (float*)((uintptr_t)arena + o_pix);
        op_conv("esrgan.conv_last", hr, B, h, w, nf, es_last, 1, h, w, Act{}, Act{}, nullptr, 0, pix, c.out_nc);
        release(hr);
        { Op o{}; o.kind = OP_COPY_OUT; o.name = "esrgan.out"; o.p0 = pix; o.cvt_n = (size_t)B * h * w * c.out_nc * 4; ops.push_back(o); }
        if (pass == 0) { arena_peak_dry = arena_peak; arena = saved; }
    }
    pB2 = B; ph = H; pw = W; pM = 0;
    return LDX_OK;
}

int Engine::run_esrgan(const float* px, int B, int H, int W, float* out, hipStream_t st) {
    if (!finalized || kind != KIND_ESRGAN) { set_error("ldx_esrgan_forward: not a finalized ESRGAN engine"); return LDX_ESTATE; }
    if (!px || !out || B <= 0 || H <= 0 || W <= 0) { set_error("ldx_esrgan_forward: bad argument"); return LDX_EINVAL; }
    HIP_OK(hipSetDevice(device));
    if (B != pB2 || H != ph || W != pw) {
        HIP_OK(hipStreamSynchronize(st));
        int rc = plan_esrgan(B, H, W);
        if (rc) return rc;
    }
    b_x = px; b_out = out; prof_graph = false;
    int rc = exec_ops(st);
    if (rc) return rc;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error(std::string("kernel launch: ") + hipGetErrorString(e)); return LDX_EHIP; }
    return LDX_OK;
}

}  // namespace ldx
