// Patch-resident 3x3 convolution for narrow outputs (round 5): N = 32 / 64 / 128 output channels (and the 3 / 4-channel tails conv_last / conv_out / the UNet's out conv
// as one 16-column tile whose missing weight rows read as zeros), stride 1, pad 1, optional nearest-resize gather.
//
// Why: ESRGAN's RRDBNet (UltimateSDUpscale/RDRB.py:80-205) is 345 convs of 64..192 -> 32 / 64 channels over a 512^2 tile, and the VAE decoder's last level
// (Decoder, VariationalAE.py:416-567: ResnetBlocks of AutoEncoders/ResBlock.py:341) runs 128 -> 128 convs over 1024^2 .. 2048^2 pixels.  As an implicit GEMM with 128 x 32 .. 256 x 128 tiles every K-tile
// re-fetches the A rows of ONE tap: 9 x the input through the L2 -> LDS path, 32 - 85 flop per operand byte, and those launches sit at the ~24 B/clk/CU
// that path delivers (ESRGAN 375 TFLOP/s = 26.9 ms per tile, VAE 128-channel convs 530 TFLOP/s; DESIGN r4 item 6).  Holding a whole 5-pixel-halo dense
// block in LDS does not fit — but the 9 taps of ONE conv share one input patch, and that does:
//
//   workgroup = 8 waves, persistent (one per CU), walking 32 x 16-pixel output tiles; wave w owns rows 4 (w & 3) .. + 3 of the tile's left / right half;
//   the input is streamed in 32-channel chunks: chunk kc of the 34 x 18 patch (612 pixels x 64 B = 38.25 KiB, LDS-DMA, ring of PR chunks) serves all 9 taps;
//   the weights of (chunk, kx) — 3 ky x N x 32 channels = 3 N / 16 KiB — ride a ring of RW stages, RW - 1 (chunk, kx) steps ahead;
//   per step a wave reads 6 patch-row fragments (rows 4w .. 4w+5 at column offset kx: each serves up to 3 (row, ky) pairs) and 3 N / 16 weight
//   fragments, and issues 12 N / 16 MFMA 16x16x32: 1.2 x the input + the weights once per 512 pixels instead of 9 x the input.
//   The (tile, step) stream never drains: the loads of the next tile's first chunks are issued during the current tile's last steps.
//
// Sizing: a load issued under load lands ~1.1 us later (MI355X_MICROARCH.md "ldsdma-fill"), a step is 768 (N = 32) .. 3072 (N = 128) MFMA cycles per SIMD,
// so N = 32 keeps 5 weight stages + 2 patch chunks in flight (PR 3, RW 6), N = 64: 5 + 1 (PR 2, RW 6), N = 128: 2 + 1 (PR 2, RW 3) — 150-155 KiB of LDS each.
// (The first version — 16 x 16 tiles, 4 waves, two workgroups per CU, one weight stage ahead — ran at a third of its MFMA time: ESRGAN 26.6 -> 15.8 ms.)
// What bounds it now (profiles/r05/conv_patch_ablations.txt): a CU pulls ~57 KB/us through its load path — one chunk of N = 32 (39 KB of patch + 18 KB of weights)
// per microsecond against 0.96 us of MFMA time for its three steps — plus ~4 us per tile of output stage, first-load latency and barriers.
//
// LDS layouts are chosen through the DMA's per-lane SOURCE address (the destination of a piece is always 64 consecutive 16-B slots):
//   patch slot (P = 34 Y + X, pos) holds channel octet  pos ^ 2 ((X >> 2) & 1)  — the 16 lanes of a ds_read_b128 group (pixels X = 16 half + l15 + kx,
//   octet g4) then cover 16 distinct bank quads for kx = 0, 1, 2 (brute-forced); a weight piece (ky, j) holds octet g of row 16 j + n at slot 4 n + (g ^ (-(n >> 2) & 3))
//   (same property; with one row per 16 lanes instead — slot 16 g + n — every lane of a DMA instruction touched its own cache line).
// One s_barrier per step; DMA completion by counted vmcnt: pieces retire in issue order, so "everything up to the loads of that step" is the wave's
// running issue count minus the count it recorded when it issued them (wave-uniform scalars; the loop is unrolled by 6 so ring slots are static).
// Epilogue: bias, LeakyReLU(0.2), output scale, residual, second scaled residual (RDRB.py:76, 205), 16-bit and / or fp32 stores, and for N = 128 the
// consumer GroupNorm's partial sums (4 channels per group = one lane's columns), one chunk per tile.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "gemm_common.h"

namespace ldx {

typedef __attribute__((ext_vector_type(4))) int cp_i32x4;
static __device__ __forceinline__ cp_i32x4 cp_srd(const void* base, long bytes) {
    const unsigned long long q = (unsigned long long)base;
    const int n = (int)(bytes > 0x7fffffffL ? 0x7fffffffL : (bytes > 0 ? bytes : 0));
    return (cp_i32x4){(int)(unsigned)q, (int)((unsigned)(q >> 32) & 0xffffu), n, 0x00020000};
}
// M0 is written without being declared (gemm_pp.inc explains why that is safe in these kernels)
static __device__ __forceinline__ void cp_dma16(const cp_i32x4 rsrc, int voff, int soff, unsigned lds) {
    // both are wave-uniform; under SGPR pressure the compiler keeps such values in vector registers and would hand those to the asm
    soff = __builtin_amdgcn_readfirstlane(soff);
    lds = (unsigned)__builtin_amdgcn_readfirstlane((int)lds);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(lds), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
// s_waitcnt vmcnt(n) for a wave-uniform run-time n in [0, 23] (larger: 23 — a smaller count than allowed only waits longer): a computed jump into
// a table of (s_waitcnt, s_branch) pairs.  A 20-way switch compiled to ~40 compare-and-branch instructions per step.
static __device__ __forceinline__ void cp_wait_n(int n) {
    n = __builtin_amdgcn_readfirstlane(n > 23 ? 23 : (n < 0 ? 0 : n));
    int t;
    asm volatile(
        "s_getpc_b64 vcc\n\t"                  // vcc = address of the next instruction
        "s_lshl_b32 %0, %1, 3\n\t"             // 8 bytes per table entry
        "s_add_u32 %0, %0, 20\n\t"             // 5 x 4 bytes from there to the table
        "s_add_u32 vcc_lo, vcc_lo, %0\n\t"
        "s_addc_u32 vcc_hi, vcc_hi, 0\n\t"
        "s_setpc_b64 vcc\n\t"
        "s_waitcnt vmcnt(0)\n\ts_branch .Lcpw%=\n\t"  "s_waitcnt vmcnt(1)\n\ts_branch .Lcpw%=\n\t"  "s_waitcnt vmcnt(2)\n\ts_branch .Lcpw%=\n\t"
        "s_waitcnt vmcnt(3)\n\ts_branch .Lcpw%=\n\t"  "s_waitcnt vmcnt(4)\n\ts_branch .Lcpw%=\n\t"  "s_waitcnt vmcnt(5)\n\ts_branch .Lcpw%=\n\t"
        "s_waitcnt vmcnt(6)\n\ts_branch .Lcpw%=\n\t"  "s_waitcnt vmcnt(7)\n\ts_branch .Lcpw%=\n\t"  "s_waitcnt vmcnt(8)\n\ts_branch .Lcpw%=\n\t"
        "s_waitcnt vmcnt(9)\n\ts_branch .Lcpw%=\n\t"  "s_waitcnt vmcnt(10)\n\ts_branch .Lcpw%=\n\t" "s_waitcnt vmcnt(11)\n\ts_branch .Lcpw%=\n\t"
        "s_waitcnt vmcnt(12)\n\ts_branch .Lcpw%=\n\t" "s_waitcnt vmcnt(13)\n\ts_branch .Lcpw%=\n\t" "s_waitcnt vmcnt(14)\n\ts_branch .Lcpw%=\n\t"
        "s_waitcnt vmcnt(15)\n\ts_branch .Lcpw%=\n\t" "s_waitcnt vmcnt(16)\n\ts_branch .Lcpw%=\n\t" "s_waitcnt vmcnt(17)\n\ts_branch .Lcpw%=\n\t"
        "s_waitcnt vmcnt(18)\n\ts_branch .Lcpw%=\n\t" "s_waitcnt vmcnt(19)\n\ts_branch .Lcpw%=\n\t" "s_waitcnt vmcnt(20)\n\ts_branch .Lcpw%=\n\t"
        "s_waitcnt vmcnt(21)\n\ts_branch .Lcpw%=\n\t" "s_waitcnt vmcnt(22)\n\ts_branch .Lcpw%=\n\t" "s_waitcnt vmcnt(23)\n"
        ".Lcpw%=:"
        : "=&s"(t) : "s"(n) : "vcc", "scc", "memory");
}

// Residual operands of one output row for all NJ column tiles, as UNCONDITIONAL 8-byte loads issued together (a null operand reads the weights instead and
// is dropped by a select).  Written per (row, column tile) behind `if (p.R)`, hipcc emitted load + s_waitcnt vmcnt(0) pairs: up to 32 HBM latencies in a
// row per tile (the same pattern as the split-K reduce launches) — ESRGAN's conv5 (two residuals) and the VAE's conv2 paid them on every tile.
template <typename T, int NJ>
static __device__ __forceinline__ void cp_load_residuals(const GemmArgs& p, const long m, const int g4, uint2 (&r1)[NJ], uint2 (&r2)[NJ]) {
    const char* z = (const char*)p.W;
    const char* b1 = p.R ? (const char*)((const T*)p.R + m * p.ldr + 4 * g4) : z;
    const char* b2 = p.R2 ? (const char*)((const T*)p.R2 + m * p.ldr2 + 4 * g4) : z;
    const int s1 = p.R ? 32 : 0, s2 = p.R2 ? 32 : 0;      // 16 columns = 32 bytes per column tile
#pragma unroll
    for (int j = 0; j < NJ; ++j) { r1[j] = *(const uint2*)(b1 + j * s1); r2[j] = *(const uint2*)(b2 + j * s2); }
}

constexpr int CP_TH = 16, CP_TW = 32, CP_PW = CP_TW + 2, CP_PIX = (CP_TH + 2) * CP_PW;     // 612 patch pixels
constexpr int CP_PP = (CP_PIX * 4 + 63) / 64;                                             // 39 pieces of 64 slots per channel chunk
constexpr int CP_PATCH = CP_PP * 1024;
constexpr int CP_RED = 8 * 32 * 2 * 4;                                                    // GroupNorm partials of the 8 waves
constexpr int cp_lds_bytes(int NJ, int PR, int RW) { return PR * CP_PATCH + RW * (3 * NJ) * 1024 + CP_RED; }

template <typename T, int NJ, int PR, int RW>
__global__ __launch_bounds__(512, 1) void conv_patch_kernel(const GemmArgs p, const int ntiles, const int abl) {
    constexpr int OOB = (int)0x80000000;
    constexpr int WP = 3 * NJ, WST = WP * 1024, NWP = (WP + 7) / 8, NPP = (CP_PP + 7) / 8;
    static_assert(RW % 3 == 0 && PR >= 2 && PR <= 3, "ring depths");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using V8 = typename Vec<T>::v8;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = wave >> 2, wrow = 4 * (wave & 3);
    const int l15 = lane & 15, g4 = lane >> 4;
    const int tiles_x = p.Wout / CP_TW, tiles_y = p.Hout / CP_TH, per_img = tiles_x * tiles_y;
    const int nimg = p.M / (p.Hout * p.Wout);
    const int Cin = p.Cin, nkc = Cin >> 5, nsteps = 3 * nkc;
    const int ntl = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;       // tiles of this workgroup

    const cp_i32x4 rA = cp_srd(p.A, (((long)nimg * p.Hin * p.Win - 1) * p.lda + Cin) * 2);
    const cp_i32x4 rW = cp_srd(p.W, (long)p.N * p.K * 2);
    const float rs_y = p.resize ? (float)p.Hin / (float)p.Hv : 1.f, rs_x = p.resize ? (float)p.Win / (float)p.Wv : 1.f;

    // tile it of this workgroup -> (image, y0, x0); an XCD's workgroups walk one contiguous range of tiles, neighbours at the same time (shared halos in its L2)
    auto tile_of = [&](int it, int& b, int& y0, int& x0) __attribute__((always_inline)) {
        const int lin = xcd_remap((int)blockIdx.x + it * (int)gridDim.x, ntiles);
        b = lin / per_img;
        const int rem = lin - b * per_img, ty = rem / tiles_x;
        y0 = ty * CP_TH; x0 = (rem - ty * tiles_x) * CP_TW;
    };
    // this wave's patch pieces of a tile: piece = wave + 8 i; slot q = 64 piece + lane -> pixel P = q >> 2 = 34 Y + X, position q & 3
    auto patch_offsets = [&](int it, int (&pv)[NPP]) __attribute__((always_inline)) {
        int b, y0, x0;
        tile_of(it < ntl ? it : 0, b, y0, x0);
        int ln;                              // opaque copy of the lane id: the per-piece (Y, X, octet) are recomputed here, once per tile, instead of
        asm volatile("v_mov_b32 %0, %1" : "=v"(ln) : "v"(lane));      // living (hoisted, then spilled) across the MFMA loop
#pragma unroll
        for (int i = 0; i < NPP; ++i) {
            const int q = (wave + 8 * i) * 64 + ln, P = q >> 2;
            const int Y = P / CP_PW, X = P - Y * CP_PW;
            const int oct = (q & 3) ^ (((X >> 2) & 1) << 1);
            const int vy = y0 - 1 + Y, vx = x0 - 1 + X;
            int sy = vy, sx = vx;
            if (p.resize) {      // nearest: src = min(floor(dst * in / out), in - 1)  (torch upsample_nearest; as gemm.hip)
                sy = min((int)floorf((float)vy * rs_y), p.Hin - 1);
                sx = min((int)floorf((float)vx * rs_x), p.Win - 1);
            }
            const bool in = P < CP_PIX && vy >= 0 && vy < p.Hv && vx >= 0 && vx < p.Wv;
            pv[i] = in ? (((b * p.Hin + sy) * p.Win + sx) * p.lda + oct * 8) * 2 : OOB;      // < 2^31: conv_patch_ok
        }
    };
    int pv[NPP];                         // ... of the tile whose chunks are being issued (it_in below)
    patch_offsets(0, pv);
    // this wave's weight pieces of a (chunk, kx) stage: piece pw = wave + 8 i = ky * NJ + j; lane -> row 16 j + (lane >> 2), slot position lane & 3
    int wv[NWP];
#pragma unroll
    for (int i = 0; i < NWP; ++i) {
        const int pw = wave + 8 * i, ky = pw / NJ, j = pw - ky * NJ;
        const int n = lane >> 2, oct = (lane & 3) ^ ((4 - (n >> 2)) & 3);      // 4 adjacent lanes = the 64 contiguous bytes of one weight row
        wv[i] = pw < WP ? ((j * 16 + n) * p.K + ky * 3 * Cin + oct * 8) * 2 : OOB;
    }
    const int nP = (CP_PP - wave + 7) / 8, nW = WP > wave ? (WP - wave + 7) / 8 : 0;      // pieces this wave issues per patch chunk / weight stage
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned lds_w = lds_base + PR * CP_PATCH;
    float* const red = (float*)(smem + PR * CP_PATCH + RW * WST);

    int issued = 0;                      // pieces this wave has issued so far
    int pslot_in = 0;                    // ring slot of the next patch chunk to issue
    int pq[PR];                          // `issued` right after patch chunk (the one being read) + i went out: a shift register, so every index is static
                                         // (per-ring-slot marks set under `if (slot == ..)` made the compiler move the whole bookkeeping to vector registers)
#pragma unroll
    for (int i = 0; i < PR; ++i) pq[i] = 0;
    auto issue_patch = [&](int kc) __attribute__((always_inline)) {
        const unsigned dst = lds_base + pslot_in * CP_PATCH;
#pragma unroll
        for (int i = 0; i < NPP; ++i)
            if (wave + 8 * i < CP_PP && !(abl & 1)) cp_dma16(rA, pv[i], kc * 64, dst + (wave + 8 * i) * 1024);
        issued += nP;
        pslot_in = pslot_in + 1 == PR ? 0 : pslot_in + 1;
    };
    auto issue_w = [&](int s, int slot) __attribute__((always_inline)) {      // stage of step s = 3 kc + kx of a tile
        const int kc = s / 3, kx = s - 3 * kc;
        const unsigned dst = lds_w + slot * WST;
        const int soff = (kx * Cin + kc * 32) * 2;
#pragma unroll
        for (int i = 0; i < NWP; ++i)
            if (wave + 8 * i < WP && !(abl & 2)) cp_dma16(rW, wv[i], soff, dst + (wave + 8 * i) * 1024);
        issued += nW;
    };

    // prologue of the stream: patch chunks 0 .. PR - 2 and weight stages 0 .. RW - 2 of the first tile (nkc >= 2 >= PR - 1, nsteps >= 6 >= RW: conv_patch_ok)
    int mark[RW];                        // `issued` right after the weights of the stage in ring slot u went out
#pragma unroll
    for (int c = 0; c < PR - 1; ++c) { issue_patch(c); pq[c] = issued; }
#pragma unroll
    for (int u = 0; u < RW - 1; ++u) { issue_w(u, u); mark[u] = issued; }
    mark[RW - 1] = issued;
    int kc_in = PR - 1;                  // next patch chunk to issue: chunk kc_in of tile it_in
    int it_in = 0;
    if (kc_in >= nkc) { kc_in -= nkc; it_in = 1; patch_offsets(1, pv); }
    int s_in = RW - 1, itw_in = 0;       // next weight stage to issue: step s_in of tile itw_in
    if (s_in >= nsteps) { s_in -= nsteps; itw_in = 1; }

    // fragment offsets: patch pixel (Y, X = 16 half + l15 + kx), octet g4 at slot position g4 ^ 2 ((X >> 2) & 1); weights at piece * 1024 + (4 l15 + (g4 ^ (-(l15 >> 2) & 3))) * 16
    int a_off[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int X = 16 * half + l15 + kx;
        a_off[kx] = (wrow * CP_PW + X) * 64 + ((g4 ^ (((X >> 2) & 1) << 1)) << 4);
    }
    const int w_off = (l15 * 4 + (g4 ^ ((4 - (l15 >> 2)) & 3))) * 16;
    int pslot = 0;                       // ring slot of the patch chunk being read

    // PIPE (N <= 64): the fragments of step s + 1 are read during the MFMAs of step s (register double buffer), so the barrier of step s waits for the
    // loads of step s + 1.  Without it the 8 waves run phase-locked — everybody reads (12 ds_read_b128 per wave = half the MFMA time at N = 32), then
    // everybody multiplies: 21 % MFMA utilisation (profiles/r05/pmc_conv_patch_v2_nopipe.txt).  N = 128 has no registers for it and 4 x the MFMAs per read.
    constexpr bool PIPE = NJ <= 2;
    constexpr int U = 6;                 // unroll: a multiple of RW (static ring slots), of 3 (static kx) and of 2 (static register buffer)
    static_assert(U % RW == 0, "unroll");
    V8 af[PIPE ? 2 : 1][6], wf[PIPE ? 2 : 1][PIPE ? 3 * NJ : 1];
    auto read_frags = [&](int buf, int kx, int ps, int wslot) __attribute__((always_inline)) {
        const char* pa = smem + ps * CP_PATCH + a_off[kx];
        const char* pw = smem + PR * CP_PATCH + wslot * WST + w_off;
#pragma unroll
        for (int i = 0; i < 6; ++i) af[buf][i] = as_v8<T>(*(const uint4*)(pa + i * (CP_PW * 64)));
        if constexpr (PIPE) {
#pragma unroll
            for (int q = 0; q < 3 * NJ; ++q) wf[buf][q] = as_v8<T>(*(const uint4*)(pw + q * 1024));
        }
    };
    if constexpr (PIPE) {                // step 0's loads, then its fragments
        const int m0 = mark[0] > pq[0] ? mark[0] : pq[0];
        cp_wait_n(issued - m0);
        asm volatile("s_barrier" ::: "memory");
        read_frags(0, 0, 0, 0);
    }

    for (int it = 0; it < ntl; ++it) {
        f32x4 acc[4][NJ];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[r][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

        for (int s0 = 0; s0 < nsteps; s0 += U) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int kx = u % 3, slot = u % RW, buf = PIPE ? (u & 1) : 0;
                // the step whose loads must have landed behind this barrier: PIPE: the next one (read below), else this one
                const int qslot = PIPE ? (u + 1) % RW : slot;
                const int qps = (PIPE && kx == 2) ? (pslot + 1 == PR ? 0 : pslot + 1) : pslot;
                const bool more = !PIPE || s0 + u + 1 < nsteps || it + 1 < ntl;
                if (more) {              // pieces retire in issue order: at most (issued - the later of the two marks) may still be in flight
                    const int mw = mark[qslot], mp = pq[(PIPE && kx == 2) ? 1 : 0];
                    cp_wait_n(issued - (mw > mp ? mw : mp));
                }
                // lgkmcnt(0) through the BUILTIN: the compiler's own wait-count pass must see that no LDS read is pending here, or it guards this step's
                // MFMAs (operands read during the previous step) with lgkmcnt(5..0) waits that count the reads issued below — and the prefetch overlaps nothing
                __builtin_amdgcn_s_waitcnt(0xC07F);      // gfx9 encoding: vmcnt 63, expcnt 7, lgkmcnt 0
                asm volatile("s_barrier" ::: "memory");
                // every wave is past the previous step and holds this step's fragments (PIPE): the previous step's weight slot and (at kx = 0) the
                // previous chunk's patch slot are free
                if (kx == 0 && it_in < ntl) {
                    issue_patch(kc_in);
                    pq[PR - 1] = issued;
                    if (++kc_in == nkc) { kc_in = 0; ++it_in; patch_offsets(it_in, pv); }
                }
                if (itw_in < ntl) {
                    issue_w(s_in, (slot + RW - 1) % RW);
                    if (++s_in == nsteps) { s_in = 0; ++itw_in; }
                }
                mark[(slot + RW - 1) % RW] = issued;
                if constexpr (PIPE) {
                    if (more && !(abl & 16)) read_frags(buf ^ 1, (kx + 1) % 3, qps, qslot);
                    if (!(abl & 4))
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
#pragma unroll
                            for (int j = 0; j < NJ; ++j) acc[r][j] = mfma16(wf[buf][ky * NJ + j], af[buf][r + ky], acc[r][j]);
                } else {
                    read_frags(0, kx, pslot, slot);
                    const char* pw = smem + PR * CP_PATCH + slot * WST + w_off;
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) {
                        constexpr int JB = 4;      // 128 accumulators: keep 4 weight fragments live at a time (with all 8 of a ky, or several ky's, the kernel spilled)
#pragma unroll
                        for (int j0 = 0; j0 < NJ; j0 += JB) {
                            V8 wq[JB];
#pragma unroll
                            for (int j = 0; j < JB; ++j) wq[j] = as_v8<T>(*(const uint4*)(pw + (ky * NJ + j0 + j) * 1024));
#pragma unroll
                            for (int r = 0; r < 4; ++r)
#pragma unroll
                                for (int j = 0; j < JB; ++j) acc[r][j0 + j] = mfma16(wq[j], af[0][r + ky], acc[r][j0 + j]);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
                if (kx == 2) {
                    pslot = pslot + 1 == PR ? 0 : pslot + 1;
#pragma unroll
                    for (int i = 0; i + 1 < PR; ++i) pq[i] = pq[i + 1];
                }
            }
        }

        // output stage: lane (l15, g4) holds pixel x0 + 16 half + l15 of row y0 + wrow + r, channels 16 j + 4 g4 .. + 3
        int b, y0, x0;
        tile_of(it, b, y0, x0);
        T* Cp = (T*)p.C;
        const T* Rp = (const T*)p.R;
        float gs[NJ], gq[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) { gs[j] = 0.f; gq[j] = 0.f; }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long m = ((long)b * p.Hout + y0 + wrow + r) * p.Wout + x0 + 16 * half + l15;
            uint2 q1[NJ], q2[NJ];
            cp_load_residuals<T, NJ>(p, m, g4, q1, q2);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int n = j * 16 + 4 * g4;
                float v[4] = {acc[r][j][0], acc[r][j][1], acc[r][j][2], acc[r][j][3]};
                if (p.bias) { const float4 bv = *(const float4*)(p.bias + n); v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w; }
                if (p.act == 3) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = v[q] > 0.f ? v[q] : 0.2f * v[q];
                }
                if (p.oscale != 0.f) { v[0] *= p.oscale; v[1] *= p.oscale; v[2] *= p.oscale; v[3] *= p.oscale; }
                if (Rp) { float rr[4]; unpack4<T>(q1[j], rr); v[0] += rr[0]; v[1] += rr[1]; v[2] += rr[2]; v[3] += rr[3]; }
                if (p.R2) { float rr[4]; unpack4<T>(q2[j], rr);
                            v[0] = fmaf(v[0], p.oscale2, rr[0]); v[1] = fmaf(v[1], p.oscale2, rr[1]); v[2] = fmaf(v[2], p.oscale2, rr[2]); v[3] = fmaf(v[3], p.oscale2, rr[3]); }
                if (Cp && !(abl & 8)) *(uint2*)(Cp + m * p.ldc + n) = pack4<T>(v[0], v[1], v[2], v[3]);
                if (p.Cf) *(float4*)(p.Cf + m * p.ldcf + n) = make_float4(v[0], v[1], v[2], v[3]);
                if (NJ == 8 && p.gn_partial) {      // statistics of the 16-bit values the consumer will read; this lane's 4 columns are group 4 j + g4
#pragma unroll
                    for (int q = 0; q < 4; ++q) { const float x = to_f32(from_f32<T>(v[q])); gs[j] += x; gq[j] = fmaf(x, x, gq[j]); }
                }
            }
        }
        if (NJ == 8 && p.gn_partial) {
            // fixed order: the lane's 16 values (above), DPP over the 16 pixels of a row group, then the 8 waves one after the other — deterministic, no atomics
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const float a = row16_sum(gs[j]), q = row16_sum(gq[j]);
                if (l15 == 0) { red[(wave * 32 + 4 * j + g4) * 2 + 0] = a; red[(wave * 32 + 4 * j + g4) * 2 + 1] = q; }
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // (not __syncthreads: its fence would wait for the prefetch DMAs in flight)
            if (tid < 64) {
                float a = 0.f;
#pragma unroll
                for (int w = 0; w < 8; ++w) a += red[w * 64 + tid];
                const int chunk = (y0 / CP_TH) * tiles_x + x0 / CP_TW;
                p.gn_partial[((long)b * p.gn_nchunk + chunk) * 64 + tid] = a;      // [b][chunk][group][2]
            }
            // (the next use of `red` is a whole tile — at least RW barriers — away)
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------------------
// Warp-specialised form for N = 32 / 64: 8 consumer waves (the tiling above) + a patch loader wave + a weight loader wave.
//
// Measured on the kernel above at N = 32 (profiles/r05/conv_patch_ablations.txt): with the loads, the MFMAs, the fragment reads AND the stores removed the
// launch still takes 18.6 of its 44.7 us — one barrier per 24 MFMAs (384 cycles) with ~60 scalar instructions of ring bookkeeping, DMA issue and the
// computed vmcnt wait on EVERY wave, all eight waves phase-locked so nobody fills the gap.  Here the consumers' step is: s_barrier, 12 ds_read_b128 for the
// next step, 24 MFMAs; each loader owns its vmcnt counter (so its counts are exact without a max over two rings), runs the same bookkeeping as above once
// per workgroup instead of eight times, and reaches the step's barrier while the consumers are still multiplying.
// Every wave executes the same number of s_barriers: (PIPE ? 1 : 0) + tiles x steps.
template <typename T, int NJ, int PR, int RW>
__global__ __launch_bounds__(640, 1) void conv_patch_ws_kernel(const GemmArgs p, const int ntiles, const int abl) {
    constexpr int OOB = (int)0x80000000;
    constexpr int WP = 3 * NJ, WST = WP * 1024;
    constexpr bool PIPE = NJ <= 2;
    constexpr int U = 6;
    static_assert(U % RW == 0 && PR >= 2 && PR <= 3 && WP * (RW - 1) <= 63, "ring depths (a loader's vmcnt holds 63)");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using V8 = typename Vec<T>::v8;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_x = p.Wout / CP_TW, tiles_y = p.Hout / CP_TH, per_img = tiles_x * tiles_y;
    const int nimg = p.M / (p.Hout * p.Wout);
    const int Cin = p.Cin, nkc = Cin >> 5, nsteps = 3 * nkc;
    const int ntl = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;       // tiles of this workgroup
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    auto tile_of = [&](int it, int& b, int& y0, int& x0) __attribute__((always_inline)) {
        const int lin = xcd_remap((int)blockIdx.x + it * (int)gridDim.x, ntiles);
        b = lin / per_img;
        const int rem = lin - b * per_img, ty = rem / tiles_x;
        y0 = ty * CP_TH; x0 = (rem - ty * tiles_x) * CP_TW;
    };

    if (wave == 8) {
        // ---- patch loader: chunk stream (tile, kc) in order, PR - 1 chunks ahead of the consumers; all 39 pieces of a chunk from this wave
        const cp_i32x4 rA = cp_srd(p.A, (((long)nimg * p.Hin * p.Win - 1) * p.lda + Cin) * 2);
        const float rs_y = p.resize ? (float)p.Hin / (float)p.Hv : 1.f, rs_x = p.resize ? (float)p.Win / (float)p.Wv : 1.f;
        // Per-lane source offsets (slot q = 64 piece + lane -> pixel q >> 2 = 34 Y + X, position q & 3).  Without a resize they are the sum of a part that does
        // not depend on the tile — pv[i] = (Y Win + X) lda 2 + 16 octet, computed ONCE — and the patch origin (a scalar, passed as the DMA's scalar offset); only tiles on the image
        // border need per-lane in-bounds tests, made once per such tile.  (Recomputing all 39 offsets per tile cost the loader, and through the
        // step barrier everybody, 3.7 us per tile: 51.2 -> 36.6 us on a 4-tile launch, profiles/r05/conv_patch_ablations.txt.)  With a resize gather the
        // source pixel is not separable and the offsets are recomputed per tile as before.
        int pv[CP_PP];
        int sbase = 0, ty0 = 0, tx0 = 0;
        bool border = false, pv_rel = false;
        auto patch_offsets = [&](int it) __attribute__((always_inline)) {
            int b, y0, x0;
            tile_of(it < ntl ? it : 0, b, y0, x0);
            if (!p.resize) {
                sbase = ((b * p.Hin + y0 - 1) * p.Win + x0 - 1) * p.lda * 2;      // the patch's first pixel (negative only on border tiles, which add it per lane)
                ty0 = y0 - 1; tx0 = x0 - 1;
                border = y0 == 0 || x0 == 0 || y0 + CP_TH >= p.Hv || x0 + CP_TW >= p.Wv;
                if (border || !pv_rel) {             // pv holds the tile-independent offsets (interior tiles) or a border tile's masked absolute ones: one
                    int ln;                          // array, rebuilt only on a border tile (a fifth of the tiles of a 512^2 image) and on the way back
                    asm volatile("v_mov_b32 %0, %1" : "=v"(ln) : "v"(lane));      // (opaque lane id: otherwise the tile-independent halves are hoisted and spilled)
                    int Y = 0, X = ln >> 2;
#pragma unroll
                    for (int i = 0; i < CP_PP; ++i) {
                        if (i) { X += 16; if (X >= CP_PW) { X -= CP_PW; ++Y; } }
                        const int oct = (ln & 3) ^ (((X >> 2) & 1) << 1);
                        const int rel = ((Y * p.Win + X) * p.lda + oct * 8) * 2;
                        const int vy = ty0 + Y, vx = tx0 + X;
                        const bool ok = i * 16 + (ln >> 2) < CP_PIX && (!border || (vy >= 0 && vy < p.Hv && vx >= 0 && vx < p.Wv));
                        pv[i] = ok ? (border ? rel + sbase : rel) : OOB;
                    }
                    pv_rel = !border;
                }
                return;
            }
            int ln;                          // opaque lane id: the per-piece (Y, X) chain is recomputed per tile, not hoisted out of the tile loop and spilled
            asm volatile("v_mov_b32 %0, %1" : "=v"(ln) : "v"(lane));
            int Y = 0, X = ln >> 2;          // pixel of piece 0; every piece is 16 pixels further
#pragma unroll
            for (int i = 0; i < CP_PP; ++i) {
                if (i) { X += 16; if (X >= CP_PW) { X -= CP_PW; ++Y; } }
                const int P = i * 16 + (ln >> 2);
                const int oct = (ln & 3) ^ (((X >> 2) & 1) << 1);
                const int vy = y0 - 1 + Y, vx = x0 - 1 + X;
                const int sy = min((int)floorf((float)vy * rs_y), p.Hin - 1), sx = min((int)floorf((float)vx * rs_x), p.Win - 1);
                const bool in = P < CP_PIX && vy >= 0 && vy < p.Hv && vx >= 0 && vx < p.Wv;
                pv[i] = in ? (((b * p.Hin + sy) * p.Win + sx) * p.lda + oct * 8) * 2 : OOB;
            }
        };
        patch_offsets(0);
        int issued = 0, pslot_in = 0, kc_in = 0, it_in = 0;
        int pq[PR];                      // `issued` right after chunk (the one being read) + i went out
#pragma unroll
        for (int i = 0; i < PR; ++i) pq[i] = 0;
        auto issue_chunk = [&]() __attribute__((always_inline)) {
            const unsigned dst = lds_base + pslot_in * CP_PATCH;
            if (!(abl & 1)) {
                if (p.resize) {
#pragma unroll
                    for (int i = 0; i < CP_PP; ++i) cp_dma16(rA, pv[i], kc_in * 64, dst + i * 1024);
                } else {
                    if (!border) {                           // two loops, not a test per piece: hipcc if-converted the per-piece form and every tile paid the masks
#pragma unroll
                        for (int i = 0; i < CP_PP; ++i) cp_dma16(rA, pv[i], sbase + kc_in * 64, dst + i * 1024);
                    } else {
#pragma unroll
                        for (int i = 0; i < CP_PP; ++i) cp_dma16(rA, pv[i], kc_in * 64, dst + i * 1024);
                    }
                }
            }
            issued += CP_PP;
            pslot_in = pslot_in + 1 == PR ? 0 : pslot_in + 1;
            if (++kc_in == nkc) { kc_in = 0; ++it_in; if (it_in < ntl && !(abl & 32)) patch_offsets(it_in); }
        };
#pragma unroll
        for (int c = 0; c < PR - 1; ++c) { issue_chunk(); pq[c] = issued; }      // nkc >= 2 >= PR - 1: all of tile 0
        if constexpr (PIPE) { cp_wait_n(issued - pq[0]); asm volatile("s_barrier" ::: "memory"); }
        for (int it = 0; it < ntl; ++it)
            for (int s0 = 0; s0 < nsteps; s0 += U) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int kx = u % 3;
                    const bool more = !PIPE || s0 + u + 1 < nsteps || it + 1 < ntl;
                    if (more) cp_wait_n(issued - pq[(PIPE && kx == 2) ? 1 : 0]);
                    asm volatile("s_barrier" ::: "memory");
                    if (kx == 0 && it_in < ntl) { issue_chunk(); pq[PR - 1] = issued; }
                    if (kx == 2) {
#pragma unroll
                        for (int i = 0; i + 1 < PR; ++i) pq[i] = pq[i + 1];
                    }
                }
            }
        return;
    }
    if (wave == 9) {
        // ---- weight loader: stage stream (tile, step) in order, RW - 1 steps ahead; piece pw = ky * NJ + j, lane -> row 16 j + (lane >> 2), position lane & 3
        const cp_i32x4 rW = cp_srd(p.W, (long)p.N * p.K * 2);
        int wv[WP];
#pragma unroll
        for (int pw = 0; pw < WP; ++pw) {
            const int ky = pw / NJ, j = pw - ky * NJ;
            const int n = lane >> 2, oct = (lane & 3) ^ ((4 - (n >> 2)) & 3);
            wv[pw] = ((j * 16 + n) * p.K + ky * 3 * Cin + oct * 8) * 2;
        }
        const unsigned lds_w = lds_base + PR * CP_PATCH;
        int issued = 0, s_in = 0, itw_in = 0;
        int mark[RW];
        auto issue_stage = [&](int slot) __attribute__((always_inline)) {
            const int kc = s_in / 3, kx = s_in - 3 * kc;
            const unsigned dst = lds_w + slot * WST;
            const int soff = (kx * Cin + kc * 32) * 2;
            if (!(abl & 2)) {
#pragma unroll
                for (int pw = 0; pw < WP; ++pw) cp_dma16(rW, wv[pw], soff, dst + pw * 1024);
            }
            issued += WP;
            if (++s_in == nsteps) { s_in = 0; ++itw_in; }
        };
#pragma unroll
        for (int u = 0; u < RW - 1; ++u) { issue_stage(u); mark[u] = issued; }      // nsteps >= 6 >= RW - 1: all of tile 0
        mark[RW - 1] = issued;
        if constexpr (PIPE) { cp_wait_n(issued - mark[0]); asm volatile("s_barrier" ::: "memory"); }
        for (int it = 0; it < ntl; ++it)
            for (int s0 = 0; s0 < nsteps; s0 += U) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int slot = u % RW, qslot = PIPE ? (u + 1) % RW : slot;
                    const bool more = !PIPE || s0 + u + 1 < nsteps || it + 1 < ntl;
                    if (more) cp_wait_n(issued - mark[qslot]);
                    asm volatile("s_barrier" ::: "memory");
                    if (itw_in < ntl) issue_stage((slot + RW - 1) % RW);
                    mark[(slot + RW - 1) % RW] = issued;
                }
            }
        return;
    }

    // ---- consumers
    const int half = wave >> 2, wrow = 4 * (wave & 3);
    const int l15 = lane & 15, g4 = lane >> 4;
    int a_off[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int X = 16 * half + l15 + kx;
        a_off[kx] = (wrow * CP_PW + X) * 64 + ((g4 ^ (((X >> 2) & 1) << 1)) << 4);
    }
    const int w_off = (l15 * 4 + (g4 ^ ((4 - (l15 >> 2)) & 3))) * 16;
    int pslot = 0;
    V8 af[PIPE ? 2 : 1][6], wf[PIPE ? 2 : 1][3 * NJ];
    auto read_frags = [&](int buf, int kx, int ps, int wslot) __attribute__((always_inline)) {
        const char* pa = smem + ps * CP_PATCH + a_off[kx];
        const char* pw = smem + PR * CP_PATCH + wslot * WST + w_off;
#pragma unroll
        for (int i = 0; i < 6; ++i) af[buf][i] = as_v8<T>(*(const uint4*)(pa + i * (CP_PW * 64)));
#pragma unroll
        for (int q = 0; q < 3 * NJ; ++q) wf[buf][q] = as_v8<T>(*(const uint4*)(pw + q * 1024));
    };
    if constexpr (PIPE) { asm volatile("s_barrier" ::: "memory"); read_frags(0, 0, 0, 0); }

    for (int it = 0; it < ntl; ++it) {
        f32x4 acc[4][NJ];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[r][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int s0 = 0; s0 < nsteps; s0 += U) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int kx = u % 3, slot = u % RW, buf = PIPE ? (u & 1) : 0;
                __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0), visible to the compiler's wait-count pass (see the kernel above)
                asm volatile("s_barrier" ::: "memory");
                if constexpr (PIPE) {
                    const bool more = s0 + u + 1 < nsteps || it + 1 < ntl;
                    const int qps = kx == 2 ? (pslot + 1 == PR ? 0 : pslot + 1) : pslot;
                    if (more && !(abl & 16)) read_frags(buf ^ 1, (kx + 1) % 3, qps, (u + 1) % RW);
                } else {
                    read_frags(0, kx, pslot, slot);
                }
                if (!(abl & 4)) {
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
#pragma unroll
                            for (int j = 0; j < NJ; ++j) acc[r][j] = mfma16(wf[buf][ky * NJ + j], af[buf][r + ky], acc[r][j]);
                }
                if (kx == 2) pslot = pslot + 1 == PR ? 0 : pslot + 1;
            }
        }
        // output stage: lane (l15, g4) holds pixel x0 + 16 half + l15 of row y0 + wrow + r, channels 16 j + 4 g4 .. + 3
        int b, y0, x0;
        tile_of(it, b, y0, x0);
        T* Cp = (T*)p.C;
        const T* Rp = (const T*)p.R;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long m = ((long)b * p.Hout + y0 + wrow + r) * p.Wout + x0 + 16 * half + l15;
            uint2 q1[NJ], q2[NJ];
            if constexpr (NJ > 1) cp_load_residuals<T, NJ>(p, m, g4, q1, q2);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int n = j * 16 + 4 * g4;
                float v[4] = {acc[r][j][0], acc[r][j][1], acc[r][j][2], acc[r][j][3]};
                if constexpr (NJ == 1) {     // N <= 16 (conv_last / conv_out with 3 or 4 channels: weight rows >= N read as zeros): element-wise tail, no residuals
                    if (n < p.N) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            if (n + q < p.N) {
                                float x = v[q] + (p.bias ? p.bias[n + q] : 0.f);
                                if (p.act == 3) x = x > 0.f ? x : 0.2f * x;
                                if (p.oscale != 0.f) x *= p.oscale;
                                if (Cp && !(abl & 8)) Cp[m * p.ldc + n + q] = from_f32<T>(x);
                                if (p.Cf) p.Cf[m * p.ldcf + n + q] = x;
                            }
                        }
                    }
                    continue;
                }
                if (p.bias) { const float4 bv = *(const float4*)(p.bias + n); v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w; }
                if (p.act == 3) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = v[q] > 0.f ? v[q] : 0.2f * v[q];
                }
                if (p.oscale != 0.f) { v[0] *= p.oscale; v[1] *= p.oscale; v[2] *= p.oscale; v[3] *= p.oscale; }
                if (Rp) { float rr[4]; unpack4<T>(q1[j], rr); v[0] += rr[0]; v[1] += rr[1]; v[2] += rr[2]; v[3] += rr[3]; }
                if (p.R2) { float rr[4]; unpack4<T>(q2[j], rr);
                            v[0] = fmaf(v[0], p.oscale2, rr[0]); v[1] = fmaf(v[1], p.oscale2, rr[1]); v[2] = fmaf(v[2], p.oscale2, rr[2]); v[3] = fmaf(v[3], p.oscale2, rr[3]); }
                if (Cp && !(abl & 8)) *(uint2*)(Cp + m * p.ldc + n) = pack4<T>(v[0], v[1], v[2], v[3]);
                if (p.Cf) *(float4*)(p.Cf + m * p.ldcf + n) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
    }
}

// Which (PR, RW) the launcher instantiates per N, and the LDS they need
static bool cp_ws_enabled() { static const bool ws = !(getenv("LDX_CONV_PATCH_WS") && atoi(getenv("LDX_CONV_PATCH_WS")) == 0); return ws; }      // 0: every wave loads and multiplies (the kernel N = 128 uses)
static bool cp_disabled() { static const bool off = getenv("LDX_CONV_PATCH") && atoi(getenv("LDX_CONV_PATCH")) == 0; return off; }

bool conv_patch_ok(const GemmArgs& a) {
    if (a.dup_rows) return false;          // the dual store of a shared CFG prefix lives in the general output stage (gemm_common.h)
    if (cp_disabled() || a.mode != 1 || a.stride != 1 || a.A2 || a.pad0 || a.f8 || a.C8 || a.ln_c1 || a.geglu || a.rowvec || a.gate || a.splitk > 1) return false;
    const bool narrow = a.N <= 16;      // conv_last / conv_out: 3 or 4 output channels, element-wise epilogue (no residuals), loader-wave kernel with one column tile
    if (!narrow && a.N != 32 && a.N != 64 && a.N != 128) return false;
    if (narrow && (a.R || a.R2 || a.gn_partial || !cp_ws_enabled())) return false;
    if (a.act != 0 && a.act != 3) return false;
    if (a.Cin % 64 || a.Cin < 64 || a.K != 9 * a.Cin || a.Hout % CP_TH || a.Wout % CP_TW || a.Hv != a.Hout || a.Wv != a.Wout) return false;      // Cin % 64: 3 Cin / 32 steps in groups of 6
    if (a.lda % 8 || (!narrow && ((a.C && a.ldc % 4) || (a.R && a.ldr % 4) || (a.R2 && a.ldr2 % 4) || (a.Cf && a.ldcf % 4)))) return false;
    const long nimg = a.M / ((long)a.Hout * a.Wout);
    if (nimg * a.Hin * a.Win * a.lda * 2 >= 0x7fffffffL) return false;      // 32-bit buffer offsets
    const long ntiles = nimg * (a.Hout / CP_TH) * (a.Wout / CP_TW);
    if (a.gn_partial && (a.N != 128 || a.gn_cpg != 4 || a.gn_G != 32 || a.gn_hw != a.Hout * a.Wout || a.gn_nchunk != (a.Hout / CP_TH) * (a.Wout / CP_TW))) return false;
    return ntiles >= (narrow ? 64 : 256);      // at least one tile per CU; the 3 / 4-channel tails win from 64 tiles on (UNet out conv at 128^2 x 2: 39 -> see profiles/r05)
}
// gemm_gn_fuse's question for this kernel: chunks per image if the epilogue can produce the consumer GroupNorm's statistics, else 0
int conv_patch_gn_chunks(const GemmArgs& a, int HW, int G) {
    GemmArgs t = a; t.gn_partial = nullptr;
    if (!conv_patch_ok(t) || a.N != 128 || G != 32 || HW != a.Hout * a.Wout) return 0;
    return (a.Hout / CP_TH) * (a.Wout / CP_TW);
}

template <typename T, int NJ, int PR, int RW>
static void launch_conv_patch_t(const GemmArgs& a, hipStream_t s) {
    static DevOnce once;
    constexpr int lds = cp_lds_bytes(NJ, PR, RW);
    static_assert(lds <= 160 * 1024, "LDS");
    set_dyn_lds(once, (const void*)conv_patch_kernel<T, NJ, PR, RW>, lds);
    const int ntiles = (a.M / (a.Hout * a.Wout)) * (a.Hout / CP_TH) * (a.Wout / CP_TW);
    static const int ncu = [] { int dev = 0, n = 256; if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 0 ? n : 256; }();
    const int grid = ntiles < ncu ? ntiles : ncu;
    static const int abl = getenv("LDX_CP_ABL") ? atoi(getenv("LDX_CP_ABL")) : 0;      // timing ablations (wrong results): 1 no patch loads, 2 no weight loads, 4 no MFMAs (N = 32), 8 no stores, 16 no fragment reads (N = 32), 32 no per-tile offset recomputation (loader-wave kernel)
    hipLaunchKernelGGL((conv_patch_kernel<T, NJ, PR, RW>), dim3(grid), dim3(512), lds, s, a, ntiles, abl);
}

template <typename T, int NJ, int PR, int RW>
static void launch_conv_patch_ws(const GemmArgs& a, hipStream_t s) {
    static DevOnce once;
    constexpr int lds = cp_lds_bytes(NJ, PR, RW);
    static_assert(lds <= 160 * 1024, "LDS");
    set_dyn_lds(once, (const void*)conv_patch_ws_kernel<T, NJ, PR, RW>, lds);
    const int ntiles = (a.M / (a.Hout * a.Wout)) * (a.Hout / CP_TH) * (a.Wout / CP_TW);
    static const int ncu = [] { int dev = 0, n = 256; if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 0 ? n : 256; }();
    static const int abl = getenv("LDX_CP_ABL") ? atoi(getenv("LDX_CP_ABL")) : 0;
    hipLaunchKernelGGL((conv_patch_ws_kernel<T, NJ, PR, RW>), dim3(ntiles < ncu ? ntiles : ncu), dim3(640), lds, s, a, ntiles, abl);
}

template <typename T>
static void launch_conv_patch_n(const GemmArgs& a, hipStream_t s) {
    const bool ws = cp_ws_enabled();
    if (a.N <= 16) launch_conv_patch_ws<T, 1, 3, 6>(a, s);
    else if (a.N == 32) { if (ws) launch_conv_patch_ws<T, 2, 3, 6>(a, s); else launch_conv_patch_t<T, 2, 3, 6>(a, s); }
    else if (a.N == 64) { if (ws) launch_conv_patch_ws<T, 4, 2, 6>(a, s); else launch_conv_patch_t<T, 4, 2, 6>(a, s); }
    else launch_conv_patch_t<T, 8, 2, 3>(a, s);
}
void launch_conv_patch(const GemmArgs& a, DType dt, hipStream_t s) {
    if (dt == DT_BF16) launch_conv_patch_n<__bf16>(a, s); else launch_conv_patch_n<_Float16>(a, s);
}

}  // namespace ldx
