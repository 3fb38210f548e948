// MFMA GEMM / implicit-GEMM 3x3 convolution for gfx950.
//
// Replaces, on the reference hot path, every F.linear (cond/cast.py:107) and conv (cast.py:174)
// of UNetModel1 / VAE Decoder / CLIP: Linear, Conv2d 1x1, Conv2d 3x3 (stride 1|2, pad 1) and
// Upsample1's nearest-resize + conv, with the bias / time-embedding add / GEGLU / residual-add
// epilogues of ResBlock1._forward (ResBlock.py:315-335), BasicTransformerBlock._forward
// (transformer.py:186-245) and SpatialTransformer.forward (transformer.py:342-377) fused in.
//
// Tiling: 128x128 output tile per 256-thread workgroup (4 waves, 2x2, 64x64 per wave as 4x4
// MFMA 16x16x32 tiles), BK = 64, double-buffered LDS (2 x 32 KiB), register-staged global loads
// issued one K-tile ahead (cdna guide T14) so the HBM/L2 latency hides under 32 MFMAs per wave.
// LDS rows are 128 B (64 x 16-bit) with a 16-B-chunk XOR swizzle (chunk ^= row & 7), which makes
// both the ds_write_b128 staging and the ds_read_b128 fragment reads bank-conflict free.
// The MFMA is issued as D = W_frag x A_frag so each lane ends up with 4 consecutive output
// channels of one output row -> 8-byte coalesced epilogue stores into the NHWC activation.
#include <stdlib.h>
#include "gemm_common.h"

namespace ldx {


// BM x BN workgroup tile, 4 waves as 2x2, each wave (BM/2) x (BN/2) = MI x NJ MFMA tiles of 16x16.
// 128x128 / 128x160 for large problems; 64x64 for short-K problems whose 128-wide tiling would leave most CUs
// idle (they are latency-bound: 4-5x more, smaller workgroups hide the HBM/L2 latency with thread-level parallelism).
// WM = waves along M (2: 4 waves / 256 threads, two workgroups per CU; 4: 8 waves / 512 threads with a 256-row tile, one
// workgroup per CU — 25 % fewer L2->LDS bytes per flop for the large-M problems that are bound by that traffic).
// F8: A / W are MX fp8 bytes (GemmArgs::f8): a K-tile is still 128 B per row = 128 elements, one 16x16x128 block-scaled MFMA
// per (i, j) and K-tile instead of two 16x16x32; the E8M0 scales bypass LDS (one dword per row and K-tile, prefetched with
// the tile).  Same LDS layout, staging, split-K and epilogue as the 16-bit kernel.
template <typename T, int MODE, int BM, int BN, int WM, bool F8, bool LNF = false>
__device__ __forceinline__ void gemm_tile_body(const GemmArgs& p, const int block) {
    static_assert(!F8 || MODE == 0, "MX fp8 operands: plain GEMM only");
    static_assert(!LNF || (MODE == 0 && !F8), "folded LayerNorm: plain 16-bit GEMM only");
    constexpr int ES = F8 ? 1 : 2;              // bytes per A / W element
    constexpr int KE = 128 / ES;                // elements per K-tile (= BK for 16-bit)
    constexpr int CE = 16 / ES;                 // elements per 16-B staging chunk
    constexpr int NT = WM * 128;                // threads
    constexpr int RPP = NT / 8;                 // tile rows staged per pass (8 threads x 16 B per 128-B row)
    constexpr int LA = BM / RPP, LB = (BN + RPP - 1) / RPP;   // staging chunks per thread (A rows, W rows; BN = 160 at 64 rows per pass is ragged)
    constexpr int MI = BM / WM / 16;            // 16-row MFMA tiles per wave
    constexpr int NJ = BN / 2 / 16;             // 16-column MFMA tiles per wave
    constexpr int STAGE_BYTES = stage_bytes<BM, BN>();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using V8 = typename Vec<T>::v8;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;     // wm in [0, WM)
    const int l15 = lane & 15, g4 = lane >> 4;

    const int tiles_n = (p.N + BN - 1) / BN;   // BN = template tile width
    const int tiles_m = (p.M + BM - 1) / BM;
    const int ntiles = tiles_m * tiles_n;
    const int S = p.splitk > 1 ? p.splitk : 1;
    const int lin = xcd_remap(block, ntiles * S);
    const int bid = lin % ntiles, split = lin / ntiles;      // same-split tiles adjacent: neighbours share panels
    const int tm = bid / tiles_n, tn = bid % tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int nk_all = (p.K + KE - 1) / KE;      // K % 8 == 0; a ragged last K-tile is zero-filled by the loader
    const int kt_begin = (int)((long)split * nk_all / S), kt_end = (int)((long)(split + 1) * nk_all / S);
    const int nk = kt_end - kt_begin;

    // ---- per-thread staging coordinates: 4 A chunks + NJ W chunks of 16 B per K-tile ----
    // Loads go through buffer descriptors (buffer_load_dwordx4 ... offen with an SGPR K offset): the per-lane
    // byte offsets are loop-invariant (plain GEMM) or change only when the 3x3 tap changes (conv), and rows
    // outside M / N / the zero-padding halo use an out-of-range offset, which the hardware returns as 0 —
    // no exec-mask branches, no 64-bit address arithmetic in the K loop.
    const int srow = tid >> 3;        // 0..RPP-1 (+RPP*j)
    const int schunk = tid & 7;       // 16-B chunk within the 128-B K-slice
    constexpr int OOB = (int)0x80000000;
    const char* __restrict__ Ap = (const char*)p.A;
    const char* __restrict__ Wp = (const char*)p.W;
    const int hw = (MODE == 1) ? p.Hout * p.Wout : 1;
    const int b0 = (MODE == 1) ? m0 / hw : 0;
    const long img = (long)p.Hin * p.Win * p.lda;                       // conv: elements per input image
    // conv: the buffer window starts at the first input row this tile can touch, so that in-window byte offsets
    // stay far below 2^31 even when one image is >= 2 GiB (VAE decode at 2048^2: 2048*2048*256 ch * 2 B)
    int row0 = 0;
    if (MODE == 1) {
        const int oy0 = (m0 - b0 * hw) / p.Wout;
        row0 = max(oy0 * p.stride - (p.pad0 ? 0 : 1), 0);
        if (p.resize) row0 = min((int)floorf((float)row0 * ((float)p.Hin / (float)p.Hv)), p.Hin - 1);
    }
    const long a_base = (MODE == 0) ? (long)m0 * p.lda : (long)b0 * img + (long)row0 * p.Win * p.lda;
    const long a_total = (MODE == 0) ? (long)(p.M - 1) * p.lda + p.K : ((long)(p.M / hw) * p.Hin * p.Win - 1) * p.lda + p.Cin;
    const long a_rem = (a_total - a_base) * ES;
    const long w_rem = ((long)p.N * p.K - (long)n0 * p.K) * ES;
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)(Ap + a_base * ES), 0, (int)(a_rem > 0x7fffffffL ? 0x7fffffffL : a_rem), 0x00020000);
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)(Wp + (long)n0 * p.K * ES), 0, (int)(w_rem > 0x7fffffffL ? 0x7fffffffL : w_rem), 0x00020000);

    int a_voff[LA];                   // plain: final byte offset ; conv: per-tap byte offset (recomputed per tap)
    int a_pix[LA];                    // conv: byte offset of this row's batch image + chunk
    int a_oy[LA], a_ox[LA];
#pragma unroll
    for (int j = 0; j < LA; ++j) {
        const int m = m0 + srow + RPP * j;
        const bool ok = m < p.M;
        if (MODE == 0) {
            a_voff[j] = ok ? ((srow + RPP * j) * p.lda + schunk * CE) * ES : OOB;
            a_pix[j] = a_oy[j] = a_ox[j] = 0;
        } else {
            const int b = m / hw, rem = m - b * hw;
            const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
            a_pix[j] = ok ? (int)(((long)(b - b0) * img + schunk * 8) * 2) : OOB;
            a_oy[j] = oy * p.stride - (p.pad0 ? 0 : 1);
            a_ox[j] = ox * p.stride - (p.pad0 ? 0 : 1);
            a_voff[j] = OOB;
        }
    }
    // second conv input (GemmArgs::A2): always a defined descriptor (0 records without A2) — a conditionally initialised one sent
    // the whole conv instantiation through scratch (12 B / lane, ESRGAN 26.8 -> 33.8 ms)
    const bool has_a2 = (MODE == 1) && p.A2 != nullptr;
    const long rem2 = has_a2 ? ((long)(p.M - m0 - 1) * p.lda2 + p.Cin2) * 2 : 0;
    const __amdgpu_buffer_rsrc_t rA2 = __builtin_amdgcn_make_buffer_rsrc(
        has_a2 ? (void*)((const char*)p.A2 + (long)m0 * p.lda2 * 2) : (void*)p.A, 0, (int)(rem2 > 0x7fffffffL ? 0x7fffffffL : (rem2 > 0 ? rem2 : 0)), 0x00020000);
    int a2_voff[(MODE == 1) ? LA : 1];
#pragma unroll
    for (int j = 0; j < ((MODE == 1) ? LA : 1); ++j)
        a2_voff[j] = (has_a2 && m0 + srow + RPP * j < p.M) ? ((srow + RPP * j) * p.lda2 + schunk * 8) * 2 : OOB;
    int w_voff[LB];
#pragma unroll
    for (int j = 0; j < LB; ++j) {
        const int n = n0 + srow + RPP * j;
        w_voff[j] = (n < p.N && srow + RPP * j < BN) ? ((srow + RPP * j) * p.K + schunk * CE) * ES : OOB;
    }
    const float rs_y = (MODE == 1 && p.resize) ? (float)p.Hin / (float)p.Hv : 1.f;
    const float rs_x = (MODE == 1 && p.resize) ? (float)p.Win / (float)p.Wv : 1.f;

    uint4 ra[LA], rb[LB];
    // MX scales: lane (l15, g4) of tile i needs byte g4 of dword SA[kt][m0 + wm*(BM/WM) + 16 i + l15] (same for W rows)
    uint32_t rsa[F8 ? MI : 1], rsb[F8 ? NJ : 1];
    int sa_voff[F8 ? MI : 1], sb_voff[F8 ? NJ : 1];
    __amdgpu_buffer_rsrc_t rSA, rSW;
    if (F8) {
        const long sa_rem = ((long)nk_all * p.sa_ld - m0) * 4, sw_rem = ((long)nk_all * p.sw_ld - n0) * 4;
        rSA = __builtin_amdgcn_make_buffer_rsrc((void*)(p.SA + m0), 0, (int)(sa_rem > 0x7fffffffL ? 0x7fffffffL : sa_rem), 0x00020000);
        rSW = __builtin_amdgcn_make_buffer_rsrc((void*)(p.SW + n0), 0, (int)(sw_rem > 0x7fffffffL ? 0x7fffffffL : sw_rem), 0x00020000);
#pragma unroll
        for (int i = 0; i < MI; ++i) { const int r = wm * (BM / WM) + i * 16 + l15; sa_voff[i] = (m0 + r < p.M) ? r * 4 : OOB; }
#pragma unroll
        for (int j = 0; j < NJ; ++j) { const int r = wn * (BN / 2) + j * 16 + l15; sb_voff[j] = (n0 + r < p.N) ? r * 4 : OOB; }
    }
    int st_ky = 0, st_kx = 0, st_ci = 0;   // conv: gload() is called with kt = 0,1,2,... in order
    bool st_new_tap = true;
    if (MODE == 1 && kt_begin > 0) {
        const int k0 = kt_begin * BK, tap = k0 / p.Cin;
        if (tap >= 9) { st_ky = 3; st_kx = 0; st_ci = k0 - 9 * p.Cin; }      // inside the second-input segment
        else { st_ci = k0 - tap * p.Cin; st_ky = tap / 3; st_kx = tap - st_ky * 3; }
    }
    auto ld128 = [](const __amdgpu_buffer_rsrc_t& r, int voff, int soff) {
        const auto v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
        return make_uint4(v[0], v[1], v[2], v[3]);
    };
    auto gload = [&](int kt) __attribute__((always_inline)) {
        const int k0 = (kt_begin + kt) * KE;
        // ragged last K-tile (plain GEMM only; conv has K = 9*Cin, Cin % 64 == 0): chunks past K read as 0
        const bool kdead = (k0 + KE > p.K) && (k0 + schunk * CE >= p.K);
        if (F8) {          // out-of-range rows read scale byte 0 (2^-127) against zero-filled operands
#pragma unroll
            for (int i = 0; i < MI; ++i) rsa[i] = __builtin_amdgcn_raw_buffer_load_b32(rSA, sa_voff[i], (kt_begin + kt) * p.sa_ld * 4, 0);
#pragma unroll
            for (int j = 0; j < NJ; ++j) rsb[j] = __builtin_amdgcn_raw_buffer_load_b32(rSW, sb_voff[j], (kt_begin + kt) * p.sw_ld * 4, 0);
        }
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < LA; ++j) ra[j] = ld128(rA, kdead ? OOB : a_voff[j], k0 * ES);
        } else {
            const bool seg2 = st_ky == 3;       // second input (GemmArgs::A2): a 10th "tap" over plain rows (wave-uniform)
            if (st_new_tap) {           // wave-uniform: once per (ky,kx) tap
#pragma unroll
                for (int j = 0; j < LA; ++j) {
                    int iy = a_oy[j] + st_ky, ix = a_ox[j] + st_kx;
                    const bool ok = a_pix[j] != OOB && iy >= 0 && iy < p.Hv && ix >= 0 && ix < p.Wv;
                    if (p.resize) {   // nearest: src = min(floor(dst * in/out), in-1)  (torch upsample_nearest)
                        iy = min((int)floorf((float)iy * rs_y), p.Hin - 1);
                        ix = min((int)floorf((float)ix * rs_x), p.Win - 1);
                    }
                    const int v = ok ? a_pix[j] + ((iy - row0) * p.Win + ix) * p.lda * 2 : OOB;
                    a_voff[j] = seg2 ? a2_voff[j] : v;
                }
            }
            const int soff = st_ci * 2;
            const __amdgpu_buffer_rsrc_t rr = seg2 ? rA2 : rA;
#pragma unroll
            for (int j = 0; j < LA; ++j) ra[j] = ld128(rr, a_voff[j], soff);
            st_ci += BK;
            st_new_tap = false;
            if (!seg2 && st_ci >= p.Cin) { st_ci = 0; st_new_tap = true; if (++st_kx == 3) { st_kx = 0; ++st_ky; } }
        }
#pragma unroll
        for (int j = 0; j < LB; ++j) rb[j] = ld128(rW, kdead ? OOB : w_voff[j], k0 * ES);
    };
    auto lstore = [&](int stage) __attribute__((always_inline)) {
        char* sA = smem + stage * STAGE_BYTES;
        char* sB = sA + BM * BK * 2;
#pragma unroll
        for (int j = 0; j < (MI > NJ ? MI : NJ); ++j) {
            const int row = srow + RPP * j;
            const int off = row * 128 + ((schunk ^ (row & 7)) << 4);
            if (j < LA) *(uint4*)(sA + off) = ra[j < LA ? j : 0];
            if (j < LB && (BN % RPP == 0 || row < BN)) *(uint4*)(sB + off) = rb[j < LB ? j : 0];
        }
    };

    f32x4 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // Residual operand (GemmArgs::R) fetched NOW instead of in the epilogue: its latency hides under the K loop (the short-K projections of the
    // transformer blocks spend a visible part of their time waiting for it after the last MFMA).  Only full 4-column groups of live rows are
    // prefetched, the same predicate the epilogue's vector path uses; split-K and GEGLU launches keep the late read.
    constexpr bool RPRE = !F8 && MI * NJ <= 20;
    uint2 rpre[RPRE ? MI * NJ : 1];
    const bool use_rpre = RPRE && p.R && p.splitk <= 1 && !p.geglu && !p.C8;
    if (use_rpre) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int m = m0 + wm * (BM / WM) + i * 16 + l15, n = n0 + wn * (BN / 2) + j * 16 + 4 * g4;
                rpre[i * NJ + j] = (m < p.M && n + 3 < p.N) ? *(const uint2*)((const T*)p.R + (long)m * p.ldr + n) : make_uint2(0u, 0u);
            }
    }
    f32x4 lnacc[LNF ? MI : 1];                           // folded LayerNorm: row sums (wn = 0) / sums of squares (wn = 1), see ln_exchange
#pragma unroll
    for (int i = 0; i < (LNF ? MI : 1); ++i) lnacc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

#ifndef LDX_GEMM_NO_T14
    // Staging schedule (guide T14): the registers always hold the NEXT tile's loads.  Top of iteration kt: write tile kt+1 to the
    // other LDS stage (its loads have had a whole iteration to land) and immediately re-issue the loads of tile kt+2, then compute
    // tile kt; one barrier per K-tile.  Compared with load -> compute -> write -> barrier this keeps loads in flight across the
    // barrier and overlaps the LDS write pass with the other waves' MFMAs.
    int csa[F8 ? MI : 1], csb[F8 ? NJ : 1];              // the CURRENT tile's scales, own block's byte moved to bits 0..7
    auto take_scales = [&]() {
        if (F8) {
#pragma unroll
            for (int i = 0; i < MI; ++i) csa[i] = (int)(rsa[i] >> (8 * g4));
#pragma unroll
            for (int j = 0; j < NJ; ++j) csb[j] = (int)(rsb[j] >> (8 * g4));
        }
    };
    gload(0);
    lstore(0);
    take_scales();
    if (nk > 1) gload(1);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        int nsa[F8 ? MI : 1], nsb[F8 ? NJ : 1];
        if (kt + 1 < nk) {
            lstore(cur ^ 1);
            if (F8) {
#pragma unroll
                for (int i = 0; i < MI; ++i) nsa[i] = (int)(rsa[i] >> (8 * g4));
#pragma unroll
                for (int j = 0; j < NJ; ++j) nsb[j] = (int)(rsb[j] >> (8 * g4));
            }
            if (kt + 2 < nk) gload(kt + 2);
        }
#else
    gload(0);
    lstore(0);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const bool more = (kt + 1) < nk;
        int csa[F8 ? MI : 1], csb[F8 ? NJ : 1];          // this K-tile's scales, own block's byte moved to bits 0..7
        if (F8) {
#pragma unroll
            for (int i = 0; i < MI; ++i) csa[i] = (int)(rsa[i] >> (8 * g4));
#pragma unroll
            for (int j = 0; j < NJ; ++j) csb[j] = (int)(rsb[j] >> (8 * g4));
        }
        if (more) gload(kt + 1);
#endif
        const char* sA = smem + cur * STAGE_BYTES;
        const char* sB = sA + BM * BK * 2;
        if (F8) {
            i32x8 af8[MI], bf8[NJ];
            auto frag = [&](const char* base, int row) {
                // operand registers 0-3 of lane group g hold k = 16 g + 0..15, registers 4-7 hold k = 64 + 16 g + 0..15, and the
                // scale of lane group b applies to k = 32 b .. 32 b + 31 (measured: profiles/ubench/mx_layout.hip)
                const uint4 lo = *(const uint4*)(base + row * 128 + ((g4 ^ (row & 7)) << 4));
                const uint4 hi = *(const uint4*)(base + row * 128 + (((4 + g4) ^ (row & 7)) << 4));
                return (i32x8){(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
            };
#pragma unroll
            for (int i = 0; i < MI; ++i) af8[i] = frag(sA, wm * (BM / WM) + i * 16 + l15);
#pragma unroll
            for (int j = 0; j < NJ; ++j) bf8[j] = frag(sB, wn * (BN / 2) + j * 16 + l15);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = mfma16_mx(bf8[j], af8[i], acc[i][j], csb[j], csa[i]);
        } else
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            V8 af[MI], bf[NJ];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int row = wm * (BM / WM) + i * 16 + l15;
                const int ch = (ks * 4 + g4) ^ (row & 7);
                af[i] = as_v8<T>(*(const uint4*)(sA + row * 128 + (ch << 4)));
            }
            if constexpr (LNF) {
#pragma unroll
                for (int i = 0; i < MI; ++i) lnacc[i] = mfma16(wn ? af[i] : ones_v8<T>(), af[i], lnacc[i]);
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int row = wn * (BN / 2) + j * 16 + l15;
                const int ch = (ks * 4 + g4) ^ (row & 7);
                bf[j] = as_v8<T>(*(const uint4*)(sB + row * 128 + (ch << 4)));
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = mfma16(bf[j], af[i], acc[i][j]);
        }
#ifndef LDX_GEMM_NO_T14
        if (F8 && kt + 1 < nk) {
#pragma unroll
            for (int i = 0; i < MI; ++i) csa[i] = nsa[i];
#pragma unroll
            for (int j = 0; j < NJ; ++j) csb[j] = nsb[j];
        }
#else
        if (more) lstore(cur ^ 1);
#endif
        __syncthreads();
    }

    if constexpr (LNF) {
        float ln1[MI], ln2[MI];
        ln_exchange<MI, BM>(lnacc, (float*)smem, wm * (BM / WM), wn, l15, g4, p.K, p.ln_eps, ln1, ln2);     // the K loop ended with a barrier
        gemm_epilogue<T, BM, BN, WM, MI, NJ, true>(p, acc, m0, n0, wm, wn, l15, g4, split, S, ln1, ln2);
    } else {
        gemm_epilogue<T, BM, BN, WM, MI, NJ, false, RPRE ? MI * NJ : 1, !F8>(p, acc, m0, n0, wm, wn, l15, g4, split, S, nullptr, nullptr, rpre, use_rpre);
    }
}

template <typename T, int MODE, int BM, int BN, int WM = 2, bool F8 = false, bool LNF = false>
__global__ __launch_bounds__(WM * 128, WM == 2 ? 2 : 1) void gemm_kernel(const GemmArgs p) {
    gemm_tile_body<T, MODE, BM, BN, WM, F8, LNF>(p, blockIdx.x);
}
// Two independent plain GEMMs in one launch (Flux double blocks: the 4096-row image stream and the 256-row text stream run the
// same layer shapes with different weights): the second problem's tiles are appended to the first's, so they fill the
// partly empty last round of workgroups instead of running as a launch of their own at low occupancy.
template <typename T, int BM, int BN, bool F8>
__global__ __launch_bounds__(256, 2) void gemm2_kernel(const GemmArgs a, const GemmArgs b, const int tiles_a) {
    if ((int)blockIdx.x < tiles_a) gemm_tile_body<T, 0, BM, BN, 2, F8>(a, blockIdx.x);
    else gemm_tile_body<T, 0, BM, BN, 2, F8>(b, blockIdx.x - tiles_a);
}

// The split-K reduce launches' epilogue operands for one row and 4 consecutive columns (n % 4 == 0, n + 3 < N)
struct SkOperands { float4 bias, rv, gate; float r[4], r2[4]; };
// Round 6: split in two so that the callers can request them WITH the partials (one memory round trip per row instead of two): the per-column part
// (bias / per-image vector / gate: the same for every row a thread of the GroupNorm-producing reduce owns) and the per-row residuals, raw.
template <typename T>
static __device__ __forceinline__ void sk_load_col_operands(const GemmArgs& p, const long m, const int n, const float* rv, SkOperands& o) {
    const float* zf = p.ws;                  // readable, 16-byte aligned: what a missing operand loads (and the select below drops)
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 tb = *(const float4*)(p.bias ? p.bias + n : zf);
    const float4 tr = *(const float4*)(rv ? rv + n : zf);
    const float4 tg = *(const float4*)(p.gate ? p.gate + (long)(m / p.rows_per_batch) * p.gate_ld + n : zf);
    o.bias = p.bias ? tb : z4; o.rv = rv ? tr : z4; o.gate = p.gate ? tg : make_float4(1.f, 1.f, 1.f, 1.f);
}
template <typename T>
static __device__ __forceinline__ void sk_load_row_operands(const GemmArgs& p, const long m, const int n, uint2& t1, uint2& t2) {
    const float* zf = p.ws;
    t1 = *(const uint2*)(p.R ? (const void*)((const T*)p.R + m * p.ldr + n) : (const void*)zf);
    t2 = *(const uint2*)(p.R2 ? (const void*)((const T*)p.R2 + m * p.ldr2 + n) : (const void*)zf);
}
template <typename T>
static __device__ __forceinline__ void sk_load_operands(const GemmArgs& p, const long m, const int n, const float* rv, SkOperands& o) {
    uint2 t1, t2;
    sk_load_col_operands<T>(p, m, n, rv, o);
    sk_load_row_operands<T>(p, m, n, t1, t2);
    unpack4<T>(t1, o.r); unpack4<T>(t2, o.r2);
}
// v -> the stored values, in gemm_epilogue's operation order
template <typename T>
static __device__ __forceinline__ void sk_apply_operands(const GemmArgs& p, const SkOperands& o, float (&v)[4]) {
    const float bb[4] = {o.bias.x, o.bias.y, o.bias.z, o.bias.w}, rr[4] = {o.rv.x, o.rv.y, o.rv.z, o.rv.w}, gg[4] = {o.gate.x, o.gate.y, o.gate.z, o.gate.w};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float x = v[c];
        if (p.bias) x += bb[c];
        if (p.rowvec) x += rr[c];
        if (p.act == 1) x = x / (1.0f + __expf(-1.702f * x));
        else if (p.act == 2) x = gelu_tanh_f(x);
        else if (p.act == 3) x = x > 0.f ? x : 0.2f * x;
        if (p.gate) x *= gg[c];
        if (p.oscale != 0.f) x *= p.oscale;
        if (p.R) x += o.r[c];
        if (p.R2) x = fmaf(x, p.oscale2, o.r2[c]);
        v[c] = x;
    }
}

// sum the split-K partials (fixed order -> deterministic) and apply the epilogue; one thread per 4 columns
template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmArgs p) {
    const int nq = (p.N + 3) / 4;
    const long total = (long)p.M * nq;
    const T* __restrict__ Rp = (const T*)p.R;
    T* __restrict__ Cp = (T*)p.C;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int m = (int)(idx / nq), n = (int)(idx % nq) * 4;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        const bool full = n + 3 < p.N;
        const float* rv = p.rowvec ? p.rowvec + (long)(m / p.rows_per_batch) * p.rowvec_ld : nullptr;
        SkOperands o;
        uint2 o1 = make_uint2(0u, 0u), o2 = o1;
        if (full) { sk_load_col_operands<T>(p, m, n, rv, o); sk_load_row_operands<T>(p, m, n, o1, o2); }      // in flight together with the partials (round 6)
        if (full) {
            // four partial loads in flight at a time (the adds stay in split order: deterministic)
            int sidx = 0;
            const size_t stride = (size_t)p.M * p.N;
            const float* w = p.ws + (size_t)m * p.N + n;
            for (; sidx + 4 <= p.splitk; sidx += 4) {
                const float4 t0 = *(const float4*)(w + (size_t)sidx * stride), t1 = *(const float4*)(w + (size_t)(sidx + 1) * stride);
                const float4 t2 = *(const float4*)(w + (size_t)(sidx + 2) * stride), t3 = *(const float4*)(w + (size_t)(sidx + 3) * stride);
                v[0] += t0.x; v[1] += t0.y; v[2] += t0.z; v[3] += t0.w;
                v[0] += t1.x; v[1] += t1.y; v[2] += t1.z; v[3] += t1.w;
                v[0] += t2.x; v[1] += t2.y; v[2] += t2.z; v[3] += t2.w;
                v[0] += t3.x; v[1] += t3.y; v[2] += t3.z; v[3] += t3.w;
            }
            for (; sidx < p.splitk; ++sidx) { const float4 t = *(const float4*)(w + (size_t)sidx * stride); v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w; }
        } else {
            for (int sidx = 0; sidx < p.splitk; ++sidx) {
                const float* w = p.ws + ((size_t)sidx * p.M + m) * p.N + n;
                for (int r = 0; r < 4 && n + r < p.N; ++r) v[r] += w[r];
            }
        }
        if (full) {
            // Epilogue operands as UNCONDITIONAL 16 / 8-byte loads (a null operand reads the workspace instead and is discarded by a select): written as
            // `if (p.bias) x += p.bias[n + r]` per element, hipcc emitted one global load + s_waitcnt vmcnt(0) per element and operand — two dozen memory
            // latencies in a row, 13 - 31 us per reduce launch, 0.65 ms of the 1024^2 step (profiles/r05/bench_kernel_stats.csv).  Same operations, same order.
            unpack4<T>(o1, o.r); unpack4<T>(o2, o.r2);
            sk_apply_operands<T>(p, o, v);
        } else
        for (int r = 0; r < 4 && n + r < p.N; ++r) {
            float x = v[r];
            if (p.bias) x += p.bias[n + r];
            if (rv) x += rv[n + r];
            if (p.act == 1) x = x / (1.0f + __expf(-1.702f * x));
            else if (p.act == 2) x = gelu_tanh_f(x);
            else if (p.act == 3) x = x > 0.f ? x : 0.2f * x;
            if (p.gate) x *= p.gate[(long)(m / p.rows_per_batch) * p.gate_ld + n + r];
            if (p.oscale != 0.f) x *= p.oscale;
            if (Rp) x += to_f32(Rp[(long)m * p.ldr + n + r]);
            if (p.R2) x = fmaf(x, p.oscale2, to_f32(((const T*)p.R2)[(long)m * p.ldr2 + n + r]));
            v[r] = x;
        }
        if (full) {
            if (Cp) { const uint2 pk = pack4<T>(v[0], v[1], v[2], v[3]); *(uint2*)(Cp + (long)m * p.ldc + n) = pk; if (p.dup_rows) *(uint2*)(Cp + ((long)m + p.dup_rows) * p.ldc + n) = pk; }
            if (p.Cf) *(float4*)(p.Cf + (long)m * p.ldcf + n) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
            for (int r = 0; r < 4 && n + r < p.N; ++r) {
                if (Cp) { Cp[(long)m * p.ldc + n + r] = from_f32<T>(v[r]); if (p.dup_rows) Cp[((long)m + p.dup_rows) * p.ldc + n + r] = from_f32<T>(v[r]); }
                if (p.Cf) p.Cf[(long)m * p.ldcf + n + r] = v[r];
            }
        }
    }
}

// Split-K reduce that also produces the consumer GroupNorm's statistics (GemmArgs::gn_partial with splitk > 1): one workgroup per chunk of
// gn_hw / gn_nchunk rows of one image x ALL columns.  Thread (rl, quad) owns 4 fixed columns and every R-th row of the chunk, so its sums of the
// rounded outputs are per column; LDS [R][N][2]; then one thread per (group, statistic) adds its gn_cpg columns over the R row lanes in a
// fixed order (deterministic, no atomics).  Same partial layout as the GEMM epilogue's: gn_partial[b][chunk][G][2].
template <typename T>
__global__ __launch_bounds__(1024) void splitk_reduce_gn_kernel(const GemmArgs p, const int R) {
    extern __shared__ float red_gn[];
    const int nq = p.N >> 2, tid = threadIdx.x;
    const int rl = tid / nq, n = (tid - rl * nq) * 4;
    const int RB = p.gn_hw / p.gn_nchunk;
    const int b = blockIdx.y, chunk = blockIdx.x;
    const long row0 = (long)b * p.gn_hw + (long)chunk * RB;
    T* __restrict__ Cp = (T*)p.C;
    const size_t stride = (size_t)p.M * p.N;
    float gs[4] = {0.f, 0.f, 0.f, 0.f}, gq[4] = {0.f, 0.f, 0.f, 0.f};
    if (rl < R) {
        SkOperands o;                                         // per-column operands: one image per chunk, one column quad per thread -> loaded once
        sk_load_col_operands<T>(p, row0, n, p.rowvec ? p.rowvec + (long)(row0 / p.rows_per_batch) * p.rowvec_ld : nullptr, o);
        // U rows x SB splits of partial loads in flight per thread.  (Round 5: with one split at a time a chunk of R rows — one row per thread, what the
        // 16^2 / 32^2 levels get — paid one memory latency per split: 20 us for S = 11 at M = 512, as long as the GEMM in front of it.)  The adds stay in
        // split order: same bits as before.
        constexpr int U = 2, SB = 4;
        for (int r = rl; r < RB; r += R * U) {
            float v[U][4];
            bool live[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { live[u] = r + u * R < RB; v[u][0] = v[u][1] = v[u][2] = v[u][3] = 0.f; }
            uint2 o1[U], o2[U];                               // the rows' residuals, requested with the first partials (round 6: they used to follow the last partial's arrival — a second round trip per row)
#pragma unroll
            for (int u = 0; u < U; ++u) { o1[u] = o2[u] = make_uint2(0u, 0u); if (live[u]) sk_load_row_operands<T>(p, row0 + r + u * R, n, o1[u], o2[u]); }
            for (int s0 = 0; s0 < p.splitk; s0 += SB) {
                float4 t[U][SB];
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int sb = 0; sb < SB; ++sb)
                        t[u][sb] = (live[u] && s0 + sb < p.splitk) ? *(const float4*)(p.ws + (size_t)(s0 + sb) * stride + (size_t)(row0 + r + u * R) * p.N + n) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int sb = 0; sb < SB; ++sb) {
                    if (s0 + sb < p.splitk) {      // (adding the zeros of a missing split would turn a -0 sum into +0)
#pragma unroll
                        for (int u = 0; u < U; ++u) { v[u][0] += t[u][sb].x; v[u][1] += t[u][sb].y; v[u][2] += t[u][sb].z; v[u][3] += t[u][sb].w; }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (!live[u]) continue;
                const long m = row0 + r + u * R;
                unpack4<T>(o1[u], o.r); unpack4<T>(o2[u], o.r2);
                sk_apply_operands<T>(p, o, v[u]);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float xr = to_f32(from_f32<T>(v[u][c]));      // statistics of the values as stored
                    gs[c] += xr; gq[c] = fmaf(xr, xr, gq[c]);
                }
                { const uint2 pk = pack4<T>(v[u][0], v[u][1], v[u][2], v[u][3]); *(uint2*)(Cp + (long)m * p.ldc + n) = pk; if (p.dup_rows) *(uint2*)(Cp + ((long)m + p.dup_rows) * p.ldc + n) = pk; }
                if (p.Cf) *(float4*)(p.Cf + (long)m * p.ldcf + n) = make_float4(v[u][0], v[u][1], v[u][2], v[u][3]);
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) { red_gn[((long)rl * p.N + n + c) * 2] = gs[c]; red_gn[((long)rl * p.N + n + c) * 2 + 1] = gq[c]; }
    }
    __syncthreads();
    // 8 lanes per (group, statistic): lane k adds elements k, k + 8, .. of the R x cpg column sums, then a 3-step butterfly — a fixed tree, so still
    // deterministic (one thread per pair walked 80 dependent LDS reads at N = 1280: ~4 us of a 14 us launch)
    if (tid < p.gn_G * 2 * 8 && (int)blockDim.x >= p.gn_G * 2 * 8) {
        const int pr = tid >> 3, k = tid & 7, g = pr >> 1, st = pr & 1, c0 = g * p.gn_cpg, ne = R * p.gn_cpg;
        float a = 0.f;
        for (int e = k; e < ne; e += 8) { const int l = e / p.gn_cpg, c = e - l * p.gn_cpg; a += red_gn[((long)l * p.N + c0 + c) * 2 + st]; }
        a += __shfl_xor(a, 1); a += __shfl_xor(a, 2); a += __shfl_xor(a, 4);
        if (k == 0) p.gn_partial[(((long)b * p.gn_nchunk + chunk) * p.gn_G + g) * 2 + st] = a;
    } else if ((int)blockDim.x < p.gn_G * 2 * 8 && tid < p.gn_G * 2) {
        const int g = tid >> 1, st = tid & 1, c0 = g * p.gn_cpg;
        float a = 0.f;
        for (int l = 0; l < R; ++l)
            for (int c = 0; c < p.gn_cpg; ++c) a += red_gn[((long)l * p.N + c0 + c) * 2 + st];
        p.gn_partial[(((long)b * p.gn_nchunk + chunk) * p.gn_G + g) * 2 + st] = a;
    }
}

// geometry of that launch: R row lanes of N / 4 threads (<= 1024 threads, >= 2 * G), chunks of RB rows; 0 = not applicable
static int splitk_gn_geom(const GemmArgs& a, int HW, int G, int max_chunks, int* R_out) {
    if (a.N % 4 || !a.C || a.N % G || HW <= 0 || a.M % HW) return 0;
    const int nq = a.N / 4;
    if (nq > 512 || nq < 1) return 0;
    int R = nq <= 128 ? 8 : (nq <= 256 ? 4 : 2);        // a power of two (chunks of R rows must tile the image); up to 1024 threads: the kernel is latency-bound
    if (nq * R < 2 * G) return 0;
    if (gn_uses_small_kernel(a.M / HW, HW, a.N, G)) return 0;      // the one-launch small GroupNorm kernel takes this one (norm.hip): faster than apply-only

    static const int lim_env = getenv("LDX_SKGN_CHUNKS") ? atoi(getenv("LDX_SKGN_CHUNKS")) : GN_NCHUNK;      // experiment switch: chunks per image (fewer = fatter workgroups)
    int lim = max_chunks < lim_env ? max_chunks : lim_env;
    // rows per chunk: a multiple of R dividing HW, as small as keeps nchunk <= lim (more workgroups), at least R
    int RB = 0;
    for (int cand = R; cand <= HW; cand += R) if (HW % cand == 0 && HW / cand <= lim) { RB = cand; break; }
    if (!RB) return 0;
    *R_out = R;
    return HW / RB;
}

// tile selection: {BM, BN}
struct TileSel { int bm, bn; };
static const double pp_r224 = getenv("LDX_PP224_RATE") ? atof(getenv("LDX_PP224_RATE")) : 1.33, pp_r192 = getenv("LDX_PP192_RATE") ? atof(getenv("LDX_PP192_RATE")) : 1.29;   // 0: never
static inline TileSel gemm_tile(int M, int N, int K, bool geglu, int splitk, bool allow_pp = true, bool plain = false) {      // plain: a GEMM (224 / 192 wide ping-pong tiles exist), not a conv
    static const int force = getenv("LDX_GEMM_TILE") ? atoi(getenv("LDX_GEMM_TILE")) : 0;      // experiment switch: BM*1000+BN
    if (force) return {force / 1000, force % 1000};
    if (!geglu && N <= 32 && splitk <= 1) return {128, 32};          // ESRGAN dense-block convs (growth 32), 3-channel output convs
    if (!geglu && N <= 64 && splitk <= 1 && (long)((M + 127) / 128) >= 400) return {128, 64};
    // Large problems: the 256-row ping-pong kernel (gemm_pp.inc; one 8-wave workgroup per CU, LDS-DMA ring) when a simple cost model
    // says so.  Isolated rates on MI355X (profiles/kprobe.py pp, bf16): 256 x 256 tiles 1.2-1.4 PFLOP/s, 256 x 160 0.95-1.2, 256 x 128
    // ~1.0 on long K, against 0.72-0.96 for the register-staged 128-row kernel; short-K problems (a handful of K-tiles) are bound by
    // per-launch fixed costs and the output write instead and gain nothing, and N = 128 (VAE 1024^2 level) loses.
    // cost = rounds x slots x tile area / rate; LDX_PP: 0 off, 1 model (default), 2 whenever a candidate fills >= 3/4 of the CUs.
    static const int pp_policy = getenv("LDX_PP") ? atoi(getenv("LDX_PP")) : 1;
    static const int pp_mink = getenv("LDX_PP_MINK") ? atoi(getenv("LDX_PP_MINK")) : 1024;
    // GEGLU up-projections (N = 8 C, an erf per output pair in the output stage) gain from the 256-wide tiles already at K = 640: M 8192 N 5120 K 640 105 us on
    // 128 x 128, 97 on 256 x 128, 84 on 256 x 256 (profiles/r06/geglu_tiles.txt)
    if (allow_pp && pp_policy && M >= 1024 && N >= 256 && (K >= pp_mink || (geglu && K >= 640))) {
        const long S = splitk > 1 ? splitk : 1, mt = (M + 255) / 256;
        double best = 1e30; int best_bn = 0;
        const int cand[5] = {256, 224, 192, 160, 128};
        const double rate[5] = {1.35, plain ? pp_r224 : 0, plain ? pp_r192 : 0, 1.15, 0.92};
        for (int c = 0; c < 5; ++c) {
            if ((geglu && cand[c] != 128 && cand[c] != 256) || rate[c] <= 0) continue;      // GEGLU pairs value / gate columns inside 64-column slabs: wave tiles of 64 or 128 columns
            const long t = mt * ((N + cand[c] - 1) / cand[c]) * S;
            if (t < 192) continue;                                   // would leave a quarter of the CUs idle
            const double cost = (double)((t + 255) / 256) * 256.0 * 256.0 * cand[c] / rate[c];
            if (cost < best) { best = cost; best_bn = cand[c]; }
        }
        if (best_bn) {
            const int bo = geglu ? 128 : ((N % 160 == 0 && N % 128 != 0) ? 160 : 128);
            const long to = (long)((M + 127) / 128) * ((N + bo - 1) / bo) * S;
            const double cost_old = (double)((to + 511) / 512) * 512.0 * 128.0 * bo / 0.84;
            // (round 6) a ragged 128-wide ping-pong tiling (N = 320: 3 column tiles for 2.5) against 160-wide register-staged tiles that fit one per CU: measured equal
            // (M 16384 N 320 K 2880: 61.4 vs 60.0 us, profiles/r06/conv_tiles.txt), and only the exact tiling can feed the consumer GroupNorm's statistics
            const bool ragged_tie = best_bn == 128 && N % 128 != 0 && N % bo == 0 && to <= 256;
            if (pp_policy >= 2 || (best < 0.95 * cost_old && !ragged_tie)) return {256, best_bn};
        }
    }
    if (geglu) return {128, 128};
    int bn = (N % 160 == 0 && N % 128 != 0) ? 160 : 128;
    // wave quantisation: 257..511 tiles of 128x128 put two workgroups on some CUs and one on the rest (the launch takes as
    // long as the doubly-loaded CUs); if 128x160 tiles fit one per CU, every CU runs a single, 1.25x larger tile instead
    static const bool no160 = getenv("LDX_NO_TILE160") != nullptr;
    if (bn == 128 && N % 160 == 0 && splitk <= 2 && !no160) {
        const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128), t160 = (long)((M + 127) / 128) * (N / 160);
        if (t128 > 256 && t128 < 512 && t160 <= 256) bn = 160;      // with splitk == 2 (gemm_choose_splitk's one-per-CU rule): 2 x t160 <= 512 workgroups
    }
    const long tiles = (long)((M + 127) / 128) * ((N + bn - 1) / bn) * (splitk > 1 ? splitk : 1);
    // < 0.8 of one round of 2 workgroups x 256 CUs and a short K loop (latency-bound): go small.
    // Long-K problems (3x3 convs) keep the big tile: measured 112 us (128x128) vs 125 us (64x64) at 64^2, 640->640;
    // 2048x1280x1280 went 27.6 -> 18.8 us with 64x64.
    if (tiles < 400 && K <= 24 * BK) {
        // Round 5: where a 64 x 160 tiling still gives every CU work, the LDS-DMA ring kernel (gemm_ring.hip: 45.7 instead of 32 flop per operand byte AND
        // three to four K-tiles in flight) takes the launch.  (The same tile on THIS kernel's register-staged loop measured slower than 64 x 64: 25.6 vs
        // 22.8 us at 2048 x 1280 x 1280 — one K-tile in flight per CU; profiles/r05/ab_caches_tile64x160.txt.)
        if (gemm_ring_ok(M, N, K, plain, splitk)) return {64, 160};
        return {64, 64};
    }
    return {128, bn};
}

template <typename T, int MODE, int BM, int BN, int WM = 2, bool F8 = false, bool LNF = false>
static void launch_gemm_inst(const GemmArgs& a, int S, hipStream_t s) {
    gemm_gn_tile_check(a, BM, BN, S);
    const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN) * S;
    const size_t lds = 2 * stage_bytes<BM, BN>();
    static DevOnce once;
    set_dyn_lds(once, (const void*)gemm_kernel<T, MODE, BM, BN, WM, F8, LNF>, (int)lds);
    hipLaunchKernelGGL((gemm_kernel<T, MODE, BM, BN, WM, F8, LNF>), dim3(tiles), dim3(WM * 128), lds, s, a);
}

// MX fp8 operands: 256-row ping-pong tile width (160 / 128) or 0 = keep the 128-row kernel.  Same cost model as gemm_tile with the MX
// rates (isolated Flux shapes, profiles/mx_probe.py); the quantised-output epilogue needs 64-column wave tiles (BN = 128).
static int mx_pp_bn(long tiles_m256, long tiles_m128, int N, int K, int S, bool c8, long ym256 = 0, long ym128 = 0, int yN = 0) {      // y*: the second problem of a two-problem launch
    static const int pp_policy = getenv("LDX_PP") ? atoi(getenv("LDX_PP")) : 1;
    static const int pp_mink = getenv("LDX_PP_MINK_MX") ? atoi(getenv("LDX_PP_MINK_MX")) : 2048;
    if (!pp_policy || N < 256 || K < pp_mink) return 0;
    double best = 1e30; int best_bn = 0;
    // per-round rates from the 8192^3 sweep (profiles/ubench/README.md): wider tiles move fewer operand bytes per flop, and N = 3072 fits one round of 224s
    static const double r224 = getenv("LDX_MX224_RATE") ? atof(getenv("LDX_MX224_RATE")) : 2.08;      // 0: never
    static const double r192 = getenv("LDX_MX192_RATE") ? atof(getenv("LDX_MX192_RATE")) : 2.1;
    const int cand[4] = {224, 192, 160, 128};
    const double rate[4] = {r224, r192, 1.93, 1.75};
    for (int c = 0; c < 4; ++c) {
        if ((c8 && cand[c] != 128 && cand[c] != 192) || rate[c] <= 0) continue;
        const long t = tiles_m256 * ((N + cand[c] - 1) / cand[c]) * S + ym256 * ((yN + cand[c] - 1) / cand[c]);
        if (t < 192) continue;
        const double cost = (double)((t + 255) / 256) * 256.0 * 256.0 * cand[c] / rate[c];
        if (cost < best) { best = cost; best_bn = cand[c]; }
    }
    if (!best_bn) return 0;
    const long to = tiles_m128 * ((N + 127) / 128) * S + ym128 * ((yN + 127) / 128);
    const double cost_old = (double)((to + 511) / 512) * 512.0 * 128.0 * 128.0 / 1.2;
    return (pp_policy >= 2 || best < 0.95 * cost_old) ? best_bn : 0;
}

template <typename T, int MODE>
static void launch_gemm_mode(const GemmArgs& a, int S, hipStream_t s) {
    if (MODE == 0 && a.f8) {        // MX fp8 operands: a K-tile holds 128 elements, so the tile heuristics see K / 2
        if constexpr (MODE == 0) {
            static const int force = getenv("LDX_GEMM_TILE") ? atoi(getenv("LDX_GEMM_TILE")) : 0;
            const int bn = force ? (force / 1000 == 256 ? (force % 1000 == 192 ? 192 : a.C8 ? 128 : (force % 1000 == 224 ? 224 : force % 1000 == 160 ? 160 : 128)) : 0)
                                 : (a.M >= 1024 ? mx_pp_bn((a.M + 255) / 256, (a.M + 127) / 128, a.N, a.K, S, a.C8 != nullptr) : 0);
            if (bn) { launch_gemm_pp(a, bn, false, S, DTypeOf<T>::v, s); return; }
        }
        const TileSel t = gemm_tile(a.M, a.N, a.K / 2, a.geglu != 0, S, false);      // no MX ping-pong kernel yet (256 x 128 only when forced)
        if (t.bm == 256 && t.bn == 128) launch_gemm_inst<T, 0, 256, 128, 4, true>(a, S, s);      // opt-in (LDX_TILE256)
        else if (t.bm == 64) launch_gemm_inst<T, 0, 64, 64, 2, true>(a, S, s);
        else if (t.bn == 160 && !a.C8) launch_gemm_inst<T, 0, 128, 160, 2, true>(a, S, s);
        else launch_gemm_inst<T, 0, 128, 128, 2, true>(a, S, s);
        return;
    }
    const TileSel t = gemm_tile(a.M, a.N, a.K, a.geglu != 0, S, true, MODE == 0 && !a.ln_c1);
    if constexpr (MODE == 0) {
        if (a.ln_c1) {       // LayerNorm folded in (no split-K): the tile shapes a transformer block's q|k|v / q / GEGLU projections take
            if (t.bm == 256 && t.bn != 128 && !a.geglu) launch_gemm_pp(a, 160, true, 1, DTypeOf<T>::v, s);
            else if (t.bm == 256) launch_gemm_pp(a, 128, true, 1, DTypeOf<T>::v, s);
            else if (t.bm == 64) launch_gemm_inst<T, 0, 64, 64, 2, false, true>(a, 1, s);
            else if (t.bn == 160 && !a.geglu) launch_gemm_inst<T, 0, 128, 160, 2, false, true>(a, 1, s);
            else launch_gemm_inst<T, 0, 128, 128, 2, false, true>(a, 1, s);
            return;
        }
    }
    if (t.bm == 256 && t.bn > 128 && (!a.geglu || t.bn == 256)) launch_gemm_pp(a, t.bn, false, S, DTypeOf<T>::v, s);
    else if (t.bm == 256) launch_gemm_pp(a, 128, false, S, DTypeOf<T>::v, s);
    else if (t.bn == 32) launch_gemm_inst<T, MODE, 128, 32>(a, S, s);
    else if (t.bm == 128 && t.bn == 64) launch_gemm_inst<T, MODE, 128, 64>(a, S, s);
    else if (a.geglu && t.bm == 64) launch_gemm_inst<T, MODE, 128, 128>(a, S, s);      // (forced tiles only) the GEGLU pairing needs 64-column wave tiles
    else if (MODE == 0 && t.bm == 64 && t.bn == 160 && S == 1 && a.K % BK == 0) launch_gemm_ring(a, DTypeOf<T>::v, s);      // gemm_ring.hip
    else if (t.bm == 64) launch_gemm_inst<T, MODE, 64, 64>(a, S, s);
    else if (t.bn == 160) launch_gemm_inst<T, MODE, 128, 160>(a, S, s);
    else launch_gemm_inst<T, MODE, 128, 128>(a, S, s);
}

template <typename T>
static void launch_gemm_t(const GemmArgs& a0, hipStream_t s) {
    GemmArgs a = a0;
    if (!gemm_sk_fixup(a)) a.sk_count = nullptr;        // the kernels reduce in place iff sk_count is set
    const int S = (a.splitk > 1 && a.ws && !a.geglu) ? a.splitk : 1;
    if (a.mode == 0) launch_gemm_mode<T, 0>(a, S, s); else launch_gemm_mode<T, 1>(a, S, s);
    if (S > 1 && a.sk_count) return;                    // reduced by the last workgroup of every tile
    if (S > 1 && a.gn_partial) {       // reduce + GroupNorm statistics (gemm_gn_fuse set the geometry)
        int R = 1;
        const int nchunk = splitk_gn_geom(a, a.gn_hw, a.gn_G, a.gn_nchunk, &R);
        if (nchunk != a.gn_nchunk) { fprintf(stderr, "ldx: split-K GroupNorm geometry mismatch (%d vs %d)\n", nchunk, a.gn_nchunk); abort(); }
        hipLaunchKernelGGL((splitk_reduce_gn_kernel<T>), dim3(nchunk, a.M / a.gn_hw), dim3((a.N / 4) * R), (size_t)R * a.N * 2 * sizeof(float), s, a, R);
    } else if (S > 1) {
        long total = (long)a.M * ((a.N + 3) / 4);
        int grid = (int)((total + 255) / 256); if (grid > 2048) grid = 2048;
        hipLaunchKernelGGL((splitk_reduce_kernel<T>), dim3(grid), dim3(256), 0, s, a);
    }
}

bool gemm_sk_fixup(const GemmArgs& a) {
    // OPT-IN at build AND run time (-DLDX_SK_FIXUP_BUILD, LDX_SK_FIXUP=1): measured on the SD1.5 1024^2 step, same box, round 4: reduce launches 14.55 / 14.71 ms, in-kernel reduction up to
    // S = 2: 14.73 (neutral), up to S = 3 / 4: 14.69 / 14.84 (+0.13 ms) — the last arriver's S serial slab reads sit in the tail of the launch,
    // while the reduce kernel spreads the same bytes over the whole chip (and also emits the GroupNorm partials).
#ifndef LDX_SK_FIXUP_BUILD
    return false;      // the in-kernel path is not compiled in (gemm_common.h): it costs every instantiation registers
#else
    static const bool off = !(getenv("LDX_SK_FIXUP") && atoi(getenv("LDX_SK_FIXUP")) != 0);
    static const int max_s = getenv("LDX_SK_FIXUP_MAX") ? atoi(getenv("LDX_SK_FIXUP_MAX")) : SK_FIXUP_MAX_S;
    if (off || !a.sk_count || !a.ws || a.splitk < 2 || a.splitk > max_s || a.geglu) return false;
    return (long)((a.M + 63) / 64) * ((a.N + 31) / 32) <= SK_COUNTERS;      // an upper bound on the tile count of any tile shape
#endif
}

void gemm_gn_tile_check(const GemmArgs& a, int BM, int BN, int S) {
    if (!a.gn_partial || S > 1 || a.gn_cpg <= 0) return;       // split-K: the reduce launch owns the statistics and checks its own geometry
    if (a.gn_hw % BM || a.gn_nchunk != a.gn_hw / BM || BN % a.gn_cpg || (BN > 160 && !(BM == 256 && BN <= 256 && a.N % BN == 0 && a.M % BM == 0))) {      // > 160: the ping-pong tiles' column-major stage
        fprintf(stderr, "ldx: GroupNorm-statistics geometry mismatch: planned for %d chunks of an image of %d rows, %d channels per group; launching %d x %d tiles\n",
                a.gn_nchunk, a.gn_hw, a.gn_cpg, BM, BN);
        abort();
    }
}

// Planner query (GemmArgs::gn_partial): which tile will launch_gemm_mode pick for `a`, and can that tile's epilogue produce the consumer
// GroupNorm's statistics?  Mirrors launch_gemm_mode's mapping from gemm_tile() to an instantiation.  LDX_GN_FUSE=0 switches the fusion off.
int gemm_gn_fuse(GemmArgs& a, int HW, int G, int max_chunks) {
    static const bool off = getenv("LDX_GN_FUSE") && atoi(getenv("LDX_GN_FUSE")) == 0;
    if (off || a.f8 || a.C8 || a.geglu || a.ln_c1 || !a.C || G <= 0 || a.N % G || a.N % 4 || HW <= 0 || a.M % HW) return 0;
    if (a.mode == 1) {                        // conv_patch.hip takes it: one chunk per 32 x 16-pixel tile (N = 128 only), or no fusion
        GemmArgs t = a; t.gn_partial = nullptr;
        if (conv_patch_ok(t)) {
            const int nc = conv_patch_gn_chunks(a, HW, G);
            if (!nc || nc > max_chunks) return 0;
            a.gn_cpg = a.N / G; a.gn_G = G; a.gn_hw = HW; a.gn_nchunk = nc;
            return nc;
        }
    }
    const bool fix = gemm_sk_fixup(a);        // in-kernel split-K reduction: the last workgroup of a tile runs the ordinary epilogue, statistics included
    if (a.splitk > 1 && a.ws && !fix) {        // split-K with a reduce launch: that launch produces the statistics (splitk_reduce_gn_kernel)
        static const bool sk_off = getenv("LDX_GN_FUSE_SPLITK") && atoi(getenv("LDX_GN_FUSE_SPLITK")) == 0;
        int R = 1;
        const int nchunk = sk_off ? 0 : splitk_gn_geom(a, HW, G, max_chunks, &R);
        if (!nchunk) return 0;
        a.gn_cpg = a.N / G; a.gn_G = G; a.gn_hw = HW; a.gn_nchunk = nchunk;
        return nchunk;
    }
    const int cpg = a.N / G;
    const TileSel t = gemm_tile(a.M, a.N, a.K, false, fix ? a.splitk : 1, true, a.mode == 0);
    int bm, bn;
    if (t.bm == 256 && t.bn > 128) { bm = 256; bn = t.bn; }
    else if (t.bm == 256) { bm = 256; bn = 128; }
    else if (t.bn == 32) { bm = 128; bn = 32; }
    else if (t.bm == 128 && t.bn == 64) { bm = 128; bn = 64; }
    else if (t.bm == 64) { bm = 64; bn = t.bn == 160 ? 160 : 64; }
    else if (t.bn == 160) { bm = 128; bn = 160; }
    else { bm = 128; bn = 128; }
    static const bool wide_off = getenv("LDX_GN_FUSE_WIDE") && atoi(getenv("LDX_GN_FUSE_WIDE")) == 0;
    const bool wide_ok = bm == 256 && bn > 160 && bn <= 256 && !wide_off && a.N % bn == 0 && a.M % bm == 0 && !a.geglu;      // the column-major output stage of the ping-pong tiles (gemm_common.h GNW): whole tiles only
    if ((bn > 160 && !wide_ok) || bn % cpg || HW % bm || (bn / cpg) * 2 > 256) return 0;          // epilogue support (gemm_common.h GNS / GNW); groups must not straddle tile columns, tiles must not straddle images
    const int nchunk = HW / bm;
    if (nchunk > max_chunks) return 0;
    a.gn_cpg = cpg; a.gn_G = G; a.gn_hw = HW; a.gn_nchunk = nchunk;
    return nchunk;
}

// heuristic shared with the planner: how many K splits for an (M, N, K) problem
int gemm_choose_splitk(int M, int N, int K, bool geglu) {
    if (geglu) return 1;
    static const int force_sk = getenv("LDX_SPLITK") ? atoi(getenv("LDX_SPLITK")) : 0;      // experiment switch
    if (force_sk && K % BK == 0 && K / BK >= force_sk * 2) return force_sk;
    const int bn = (N % 160 == 0 && N % 128 != 0) ? 160 : 128;
    const int tiles = ((M + 127) / 128) * ((N + bn - 1) / bn);
    const int nk = K / BK;
    // one workgroup per CU (193..256 tiles, or that many 160-wide ones) runs a long K loop alone: a CU advances two co-resident
    // workgroups by one K-tile each in about the time a lone one needs for its own, so halving K for twice the workgroups nearly
    // halves the launch; the fp32 partials + reduce pass cost ~15 us (64^2-level 3x3 convs 640 -> 640: 122 -> ~80 us)
    static const bool split2 = !getenv("LDX_NO_SPLIT2");
    if (split2 && K % BK == 0 && nk >= 60) {
        const long mt = (M + 127) / 128;
        const long t160 = (N % 160 == 0) ? mt * (N / 160) : 0;
        if ((tiles > 192 && tiles <= 256) || (tiles > 256 && tiles < 512 && t160 > 192 && t160 <= 256)) return 2;
    }
    // a split costs a second (reduce) launch of ~9 us: only worth it for long K (3x3 convs, FF down-projection)
    if (tiles >= 200 || nk < 40 || K % BK) return 1;
    int s = (440 + tiles / 2) / tiles;        // ~400-480 workgroups: 40 tiles -> 11 (measured best 10), 160 tiles -> 3
    if (s > nk / 5) s = nk / 5;
    if (s > 16) s = 16;
    return s < 2 ? 1 : s;
}

template <typename T, bool F8>
static void launch_gemm2_t(const GemmArgs& a, const GemmArgs& b, hipStream_t s) {
    const int ta = ((a.M + 127) / 128) * ((a.N + 127) / 128), tb = ((b.M + 127) / 128) * ((b.N + 127) / 128);
    const size_t lds = 2 * stage_bytes<128, 128>();
    static DevOnce once;
    set_dyn_lds(once, (const void*)gemm2_kernel<T, 128, 128, F8>, (int)lds);
    hipLaunchKernelGGL((gemm2_kernel<T, 128, 128, F8>), dim3(ta + tb), dim3(256), lds, s, a, b, ta);
}
// 256-row ping-pong tiles for a two-problem launch: same cost model as gemm_tile over the combined tile count; 0 = keep 128 x 128
static int gemm2_pp_bn(const GemmArgs& a, const GemmArgs& b) {
    static const int pp_policy = getenv("LDX_PP") ? atoi(getenv("LDX_PP")) : 1;
    static const int pp_mink = getenv("LDX_PP_MINK") ? atoi(getenv("LDX_PP_MINK")) : 1024;
    if (!pp_policy || a.f8 || b.f8 || a.K < pp_mink || b.K < pp_mink || a.N < 256 || b.N < 256 || a.M + b.M < 1024) return 0;
    double best = 1e30; int best_bn = 0;
    const int cand[5] = {256, 224, 192, 160, 128};
    const double rate[5] = {1.35, pp_r224, pp_r192, 1.15, 0.92};
    for (int c = 0; c < 5; ++c) {
        if (rate[c] <= 0) continue;
        const long t = (long)((a.M + 255) / 256) * ((a.N + cand[c] - 1) / cand[c]) + (long)((b.M + 255) / 256) * ((b.N + cand[c] - 1) / cand[c]);
        if (t < 192) continue;
        const double cost = (double)((t + 255) / 256) * 256.0 * 256.0 * cand[c] / rate[c];
        if (cost < best) { best = cost; best_bn = cand[c]; }
    }
    if (!best_bn) return 0;
    const long to = (long)((a.M + 127) / 128) * ((a.N + 127) / 128) + (long)((b.M + 127) / 128) * ((b.N + 127) / 128);
    const double cost_old = (double)((to + 511) / 512) * 512.0 * 128.0 * 128.0 / 0.84;
    return (pp_policy >= 2 || best < 0.95 * cost_old) ? best_bn : 0;
}
// both plain mode, no split-K, no GEGLU, same operand kind (16-bit or MX); 128x128 tiles, or 256-row ping-pong tiles (16-bit)
void launch_gemm2(const GemmArgs& a, const GemmArgs& b, DType dt, hipStream_t s) {
    if (a.M <= 0 || a.N <= 0) { launch_gemm(b, dt, s); return; }
    if (b.M <= 0 || b.N <= 0) { launch_gemm(a, dt, s); return; }
    GemmArgs x = a, y = b; x.splitk = y.splitk = 1; { static const int epg = getenv("LDX_EP_GENERAL") ? atoi(getenv("LDX_EP_GENERAL")) : 0; x.ep_general = y.ep_general = epg; }
    if (x.f8 && y.f8 && x.M + y.M >= 1024 && y.N >= 256) {       // MX: the two streams of a Flux double block / the two halves of linear1
        const bool c8 = x.C8 || y.C8;
        const int bn = mx_pp_bn((x.M + 255) / 256, (x.M + 127) / 128, x.N, x.K < y.K ? x.K : y.K, 1, c8, (y.M + 255) / 256, (y.M + 127) / 128, y.N);
        if (bn) { launch_gemm_pp2(x, y, bn, dt, s); return; }
    }
    if (const int bn = gemm2_pp_bn(x, y)) {
        launch_gemm_pp2(x, y, bn, dt, s);
        return;
    }
    if (dt == DT_BF16) { if (a.f8) launch_gemm2_t<__bf16, true>(x, y, s); else launch_gemm2_t<__bf16, false>(x, y, s); }
    else { if (a.f8) launch_gemm2_t<_Float16, true>(x, y, s); else launch_gemm2_t<_Float16, false>(x, y, s); }
}

static const int ep_general_env = getenv("LDX_EP_GENERAL") ? atoi(getenv("LDX_EP_GENERAL")) : 0;
void launch_gemm(const GemmArgs& a0, DType dt, hipStream_t s) {
    if (a0.M <= 0 || a0.N <= 0) return;
    GemmArgs a = a0; a.ep_general = ep_general_env;
    if (a.mode == 1 && conv_patch_ok(a)) { launch_conv_patch(a, dt, s); return; }      // conv_patch.hip: narrow 3x3 convs with the input patch resident in LDS
    if (dt == DT_BF16) launch_gemm_t<__bf16>(a, s); else launch_gemm_t<_Float16>(a, s);
}

// ------------------------------------------------------------------------------------------
// Skinny GEMM (tiny M): HBM-bound on the weight matrix.  One workgroup = 4 waves x 4 output columns each; the
// (<= 4 row) activation block is staged once in LDS as fp32, and every wave keeps 4 independent 16-byte weight
// streams in flight so the loads pipeline.
template <typename T>
__global__ __launch_bounds__(256) void skinny_kernel(const SkinnyArgs p) {
    extern __shared__ float sx[];                          // [mrows][K]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n0 = (blockIdx.x * 4 + wave) * 4;
    for (int mb = 0; mb < p.M; mb += 4) {
        const int mr = min(4, p.M - mb);
        __syncthreads();
        for (int i = threadIdx.x; i < mr * p.K; i += 256) {
            float xv = p.x[(long)(mb + i / p.K) * p.ldx + (i % p.K)];
            if (p.in_act) xv = xv / (1.0f + expf(-xv));
            sx[i] = xv;
        }
        __syncthreads();
        if (n0 >= p.N) continue;
        float acc[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) acc[r][mi] = 0.f;
        const T* __restrict__ w = (const T*)p.W;
        for (int k = lane * 8; k < p.K; k += 64 * 8) {
            uint4 wv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) wv[r] = (n0 + r < p.N) ? *(const uint4*)(w + (long)(n0 + r) * p.K + k) : make_uint4(0, 0, 0, 0);
            float xr[4][8];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int e = 0; e < 8; ++e) xr[mi][e] = mi < mr ? sx[mi * p.K + k + e] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float wf[8];
                unpack8<T>(wv[r], wf);
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[r][mi] = fmaf(xr[mi][e], wf[e], acc[r][mi]);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const float v = wave_sum(acc[r][mi]);
                const int m = mb + mi, n = n0 + r;
                if (lane == 0 && mi < mr && n < p.N) {
                    float o = v + (p.bias ? p.bias[n] : 0.f);
                    if (p.out_act) o = o / (1.0f + expf(-o));
                    if (p.accum) o += p.out[(long)m * p.ldo + n];
                    p.out[(long)m * p.ldo + n] = o;
                }
            }
    }
}

// The same with MX fp8 weights (SkinnyArgs::W8): the activation block goes through the MX rule while it is staged — per 32 consecutive k: amax, E8M0 scale,
// e4m3 round-to-nearest-even, back to fp32 (oracle mx_fake_quant; the value a block-scaled MFMA would see) — and every lane streams 16 weights per 16-byte load,
// dequantised with the scale byte of their block.  fp32 accumulation.  HBM-bound like the 16-bit kernel, at half the bytes.
constexpr int SKM_RG = 4;
__global__ __launch_bounds__(256) void skinny_mx_kernel(const SkinnyArgs p) {
    extern __shared__ float sx[];                          // [mrows][K]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nkb = p.K >> 5;
    for (int mb = 0; mb < p.M; mb += 4) {
        const int mr = min(4, p.M - mb);
        __syncthreads();
        for (int blk = threadIdx.x; blk < mr * nkb; blk += 256) {      // one thread per 32-element block
            const int mi = blk / nkb, kb = blk - mi * nkb;
            const float* src = p.x + (long)(mb + mi) * p.ldx + kb * 32;
            float f[32];
            float amax = 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float4 t = *(const float4*)(src + 4 * c);
                f[4 * c] = t.x; f[4 * c + 1] = t.y; f[4 * c + 2] = t.z; f[4 * c + 3] = t.w;
            }
#pragma unroll
            for (int i = 0; i < 32; ++i) { if (p.in_act) f[i] = f[i] / (1.0f + expf(-f[i])); amax = fmaxf(amax, fabsf(f[i])); }
            const int e = mx_scale_e8m0(amax);
            const float inv = mx_inv_scale(e), sc = __uint_as_float((uint32_t)e << 23);
            float* dst = sx + (long)mi * p.K + kb * 32;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int w = (int)mx_pack4(f[4 * c] * inv, f[4 * c + 1] * inv, f[4 * c + 2] * inv, f[4 * c + 3] * inv);
                const auto lo = __builtin_amdgcn_cvt_pk_f32_fp8(w, false), hi = __builtin_amdgcn_cvt_pk_f32_fp8(w, true);
                *(float4*)(dst + 4 * c) = make_float4(lo[0] * sc, lo[1] * sc, hi[0] * sc, hi[1] * sc);
            }
        }
        __syncthreads();
        // SKM_RG groups of 4 output rows per wave: the staged (and quantised) activation serves 64 rows of W per workgroup instead of 16
        for (int rg = 0; rg < SKM_RG; ++rg) {
        const int n0 = ((blockIdx.x * 4 + wave) * SKM_RG + rg) * 4;
        if (n0 >= p.N) break;
        float acc[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) acc[r][mi] = 0.f;
        const uint8_t* __restrict__ w8 = (const uint8_t*)p.W8;
        for (int k = lane * 16; k < p.K; k += 64 * 16) {
            uint4 wv[4]; uint32_t sd[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool ok = n0 + r < p.N;
                wv[r] = ok ? *(const uint4*)(w8 + (long)(n0 + r) * p.K + k) : make_uint4(0, 0, 0, 0);
                sd[r] = ok ? p.SW[(long)(k >> 7) * p.sw_ld + n0 + r] : 0u;
            }
            float xr[4][16];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int e = 0; e < 16; ++e) xr[mi][e] = mi < mr ? sx[mi * p.K + k + e] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float sc = __uint_as_float(((sd[r] >> (8 * ((k >> 5) & 3))) & 0xffu) << 23);
                const int wq[4] = {(int)wv[r].x, (int)wv[r].y, (int)wv[r].z, (int)wv[r].w};
                float wf[16];               // the lane's 16 weights share one block: unscaled products first, the scale once per row and block
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const auto lo = __builtin_amdgcn_cvt_pk_f32_fp8(wq[c], false), hi = __builtin_amdgcn_cvt_pk_f32_fp8(wq[c], true);
                    wf[4 * c] = lo[0]; wf[4 * c + 1] = lo[1]; wf[4 * c + 2] = hi[0]; wf[4 * c + 3] = hi[1];
                }
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) {
                    if (mi < mr) {          // wave-uniform: only the live activation rows (Flux: one or two)
                        float t = 0.f;
#pragma unroll
                        for (int e = 0; e < 16; ++e) t = fmaf(xr[mi][e], wf[e], t);
                        acc[r][mi] = fmaf(t, sc, acc[r][mi]);
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const float v = wave_sum(acc[r][mi]);
                const int m = mb + mi, n = n0 + r;
                if (lane == 0 && mi < mr && n < p.N) {
                    float o = v + (p.bias ? p.bias[n] : 0.f);
                    if (p.out_act) o = o / (1.0f + expf(-o));
                    if (p.accum) o += p.out[(long)m * p.ldo + n];
                    p.out[(long)m * p.ldo + n] = o;
                }
            }
        }
    }
}

void launch_skinny(const SkinnyArgs& a, DType dt, hipStream_t s) {
    if (a.M <= 0 || a.N <= 0) return;
    dim3 grid((a.N + 15) / 16), block(256);
    const size_t lds = (size_t)4 * a.K * sizeof(float);
    if (a.W8 && (a.K & 31) == 0) {
        static DevOnce once;
        set_dyn_lds(once, (const void*)skinny_mx_kernel, 160 * 1024);
        hipLaunchKernelGGL(skinny_mx_kernel, dim3((a.N + 16 * SKM_RG - 1) / (16 * SKM_RG)), block, lds, s, a);
        return;
    }
    if (dt == DT_BF16) {
        static DevOnce once;
    set_dyn_lds(once, (const void*)skinny_kernel<__bf16>, 160 * 1024);
        hipLaunchKernelGGL((skinny_kernel<__bf16>), grid, block, lds, s, a);
    } else {
        static DevOnce once;
    set_dyn_lds(once, (const void*)skinny_kernel<_Float16>, 160 * 1024);
        hipLaunchKernelGGL((skinny_kernel<_Float16>), grid, block, lds, s, a);
    }
}

}  // namespace ldx
