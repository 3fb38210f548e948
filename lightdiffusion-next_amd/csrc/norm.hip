// GroupNorm(32)+SiLU and LayerNorm over NHWC 16-bit activations (HBM-bound kernels).
//
// Reference call sites: F.group_norm via cast.GroupNorm (cond/cast.py:224-257) in
// ResBlock1.in_layers/out_layers (ResBlock.py:251-299, eps 1e-5), SpatialTransformer.norm
// (transformer.py:286-293, eps 1e-6), UNet out (unet.py:663-677), VAE Normalize
// (Attention/Attention.py:11-31, eps 1e-6); F.layer_norm via cast.LayerNorm (cast.py:260-290)
// in BasicTransformerBlock (transformer.py:186-245) and CLIP (clip/Clip.py).
//
// NHWC makes a pixel's channels contiguous, so both kernels read/write full 16-byte chunks of
// coalesced rows.  Each thread owns a fixed 8-channel chunk (tx) and strides over pixels (ty):
// per-channel fp32 sum / sum-of-squares live in registers, are reduced through LDS in a fixed
// order (deterministic, no atomics) and written as per-(batch, pixel-chunk, group) partials.
// The apply kernel folds the partials into mean/rstd, pre-multiplies gamma/beta into one FMA per
// element and fuses SiLU.  Statistics are fp32 over the exact 16-bit inputs.
#include <stdlib.h>
#include "ldx_device.h"
#include "ldx_kernels.h"

namespace ldx {

struct GnGeom { int CH, TX, RY; };
static inline GnGeom gn_geom(int C) {
    const int cpp = C / 8;
    int ch = 1;                                   // chunks per thread: 1, 2 or 4 (template instances)
    while (ch < 4 && (cpp / ch > 256 || cpp % ch)) ch *= 2;
    GnGeom g; g.CH = ch; g.TX = cpp / ch; g.RY = 256 / g.TX; if (g.RY < 1) g.RY = 1;
    return g;
}

template <typename T, int CH>
__global__ __launch_bounds__(256) void gn_stats_kernel(const GroupNormArgs p, int TX, int RY) {
    extern __shared__ float sred[];                 // [RY][C][2]
    const int tid = threadIdx.x;
    const int tx = tid % TX, ty = tid / TX;
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int per = (p.HW + p.nchunk - 1) / p.nchunk;
    const int pb = chunk * per, pe = min(p.HW, pb + per);
    const T* __restrict__ X = (const T*)p.X + (long)b * p.HW * p.ldx;

    float s[CH][8], q[CH][8];
#pragma unroll
    for (int k = 0; k < CH; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) { s[k][e] = 0.f; q[k][e] = 0.f; }

    if (ty < RY) {
        // four pixel rows of loads in flight per thread (one at a time left the kernel latency-bound at ~2.2 TB/s); the sums still take the rows
        // in the same order as the one-row loop, so the statistics are bit-identical to it
        int pix = pb + ty;
        for (; pix + 3 * RY < pe; pix += 4 * RY) {
            uint4 u[4][CH];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int k = 0; k < CH; ++k) u[r][k] = *(const uint4*)(X + (long)(pix + r * RY) * p.ldx + (tx + TX * k) * 8);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int k = 0; k < CH; ++k) {
                    float f[8];
                    unpack8<T>(u[r][k], f);
#pragma unroll
                    for (int e = 0; e < 8; ++e) { s[k][e] += f[e]; q[k][e] = fmaf(f[e], f[e], q[k][e]); }
                }
        }
        for (; pix < pe; pix += RY) {
#pragma unroll
            for (int k = 0; k < CH; ++k) {
                float f[8];
                unpack8<T>(*(const uint4*)(X + (long)pix * p.ldx + (tx + TX * k) * 8), f);
#pragma unroll
                for (int e = 0; e < 8; ++e) { s[k][e] += f[e]; q[k][e] = fmaf(f[e], f[e], q[k][e]); }
            }
        }
#pragma unroll
        for (int k = 0; k < CH; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int c = (tx + TX * k) * 8 + e;
                sred[(ty * p.C + c) * 2 + 0] = s[k][e];
                sred[(ty * p.C + c) * 2 + 1] = q[k][e];
            }
    }
    __syncthreads();
    if (tid < p.G * 2) {
        const int g = tid >> 1, st = tid & 1;
        const int cpg = p.C / p.G;
        float acc = 0.f;
        for (int r = 0; r < RY; ++r)
            for (int c = g * cpg; c < (g + 1) * cpg; ++c) acc += sred[(r * p.C + c) * 2 + st];
        p.partial[(((long)b * p.nchunk + chunk) * p.G + g) * 2 + st] = acc;
    }
}

template <typename T, int CH>
__global__ __launch_bounds__(256) void gn_apply_kernel(const GroupNormArgs p, int TX, int RY, int nblk) {
    __shared__ float smean[64], srstd[64];
    __shared__ float sred[8][32][2];
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const int cpg = p.C / p.G;
    // Round 6: the kernel is a chain of memory latencies, not a stream — the HW 16384 x C 320 launch took 16.4 us for ONE image and for two (profiles/r06).  What
    // does not depend on the statistics is therefore requested first: the thread's first four pixel rows and its gamma / beta are in flight while the partials fold.
    const int tx = tid % TX, ty = tid / TX;
    const bool act = ty < RY;
    const T* __restrict__ X = (const T*)p.X + (long)b * p.HW * p.ldx;
    T* __restrict__ Y = (T*)p.Y + (long)b * p.HW * p.ldy;
    const int per = (p.HW + nblk - 1) / nblk;
    const int pb = blockIdx.x * per, pe = min(p.HW, pb + per);
    int pix = pb + ty;
    uint4 u0[4][CH];
    const bool first4 = act && pix + 3 * RY < pe;
    if (first4) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int k = 0; k < CH; ++k) u0[r][k] = *(const uint4*)(X + (long)(pix + r * RY) * p.ldx + (tx + TX * k) * 8);
    }
    float4 gam[CH][2], bet[CH][2];
    if (act) {
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const int c = (tx + TX * k) * 8;
            gam[k][0] = *(const float4*)(p.gamma + c); gam[k][1] = *(const float4*)(p.gamma + c + 4);
            bet[k][0] = *(const float4*)(p.beta + c);  bet[k][1] = *(const float4*)(p.beta + c + 4);
        }
    }
    {   // fold the per-chunk partials: 8 slices x 32 groups, fixed order (deterministic)
        const int g = tid & 31, sl = tid >> 5;
        float su = 0.f, sq = 0.f;
        if (g < p.G) {
            // up to 32 partial rows per thread (nchunk = 256): sixteen loads in flight at a time, added in row order (same sums as a plain loop) —
            // as a dependent load-add loop this prologue cost ~0.3 us per row and made up most of the kernel at 1024-pixel maps
            const float2* pp = (const float2*)p.partial + ((long)b * p.nchunk) * p.G + g;
            int ck = sl;
            for (; ck + 120 < p.nchunk; ck += 128) {
                float2 v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) v[u] = pp[(long)(ck + 8 * u) * p.G];
#pragma unroll
                for (int u = 0; u < 16; ++u) { su += v[u].x; sq += v[u].y; }
            }
            for (; ck + 56 < p.nchunk; ck += 64) {
                float2 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = pp[(long)(ck + 8 * u) * p.G];
#pragma unroll
                for (int u = 0; u < 8; ++u) { su += v[u].x; sq += v[u].y; }
            }
            for (; ck < p.nchunk; ck += 8) { const float2 v = pp[(long)ck * p.G]; su += v.x; sq += v.y; }
        }
        sred[sl][g][0] = su; sred[sl][g][1] = sq;
    }
    __syncthreads();
    if (tid < p.G) {
        float su = 0.f, sq = 0.f;
#pragma unroll
        for (int sl = 0; sl < 8; ++sl) { su += sred[sl][tid][0]; sq += sred[sl][tid][1]; }
        const float n = (float)p.HW * (float)cpg;
        const float mean = su / n;
        const float var = fmaxf(sq / n - mean * mean, 0.f);
        smean[tid] = mean;
        srstd[tid] = rsqrtf(var + p.eps);
    }
    __syncthreads();
    if (!act) return;
    float sc[CH][8], sh[CH][8];
#pragma unroll
    for (int k = 0; k < CH; ++k) {
        const float ga[8] = {gam[k][0].x, gam[k][0].y, gam[k][0].z, gam[k][0].w, gam[k][1].x, gam[k][1].y, gam[k][1].z, gam[k][1].w};
        const float be[8] = {bet[k][0].x, bet[k][0].y, bet[k][0].z, bet[k][0].w, bet[k][1].x, bet[k][1].y, bet[k][1].z, bet[k][1].w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = (tx + TX * k) * 8 + e;
            const int g = c / cpg;
            const float a = srstd[g] * ga[e];
            sc[k][e] = a;
            sh[k][e] = be[e] - smean[g] * a;
        }
    }
    auto emit = [&](const uint4& u, const int pix, const int k) __attribute__((always_inline)) {
        float f[8];
        unpack8<T>(u, f);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float y = fmaf(f[e], sc[k][e], sh[k][e]);
            if (p.silu) y = silu_f(y);
            f[e] = y;
        }
        *(uint4*)(Y + (long)pix * p.ldy + (tx + TX * k) * 8) = pack8<T>(f);
    };
    if (first4) {                                     // the rows requested at the top
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int k = 0; k < CH; ++k) emit(u0[r][k], pix + r * RY, k);
        pix += 4 * RY;
    }
    for (; pix + 3 * RY < pe; pix += 4 * RY) {      // four pixel rows of loads in flight per thread
        uint4 u[4][CH];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int k = 0; k < CH; ++k) u[r][k] = *(const uint4*)(X + (long)(pix + r * RY) * p.ldx + (tx + TX * k) * 8);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int k = 0; k < CH; ++k) emit(u[r][k], pix + r * RY, k);
    }
    for (; pix < pe; pix += RY) {
#pragma unroll
        for (int k = 0; k < CH; ++k) emit(*(const uint4*)(X + (long)pix * p.ldx + (tx + TX * k) * 8), pix, k);
    }
}

// Small feature maps (SD1.5 level 3: HW = 256 with C = 1280 / 2560; at HW = 1024 the two-launch path is faster): one workgroup per (group, batch) does both passes in a
// single launch — statistics, then normalise + SiLU re-reading the (L2-hot) 80..160 KB it just summed.  Requires the group's
// channels to be whole 16-byte chunks (C/G % 8 == 0).  Replaces two launches of ~9 us each whose grids were too small to matter.
template <typename T>
__global__ __launch_bounds__(256) void gn_small_kernel(const GroupNormArgs p) {
    __shared__ float red[2][4];
    const int tid = threadIdx.x, g = blockIdx.x, b = blockIdx.y;
    const int cpg = p.C / p.G, cpc = cpg >> 3;                 // channels / 16-byte chunks per group
    const long total = (long)p.HW * cpc;
    const T* __restrict__ X = (const T*)p.X + (long)b * p.HW * p.ldx + g * cpg;
    T* __restrict__ Y = (T*)p.Y + (long)b * p.HW * p.ldy + g * cpg;
    float su = 0.f, sq = 0.f;
    // Round 5: a group of the 16^2 level is 20 - 40 KB = 5 - 10 chunks per thread: ALL of them are loaded at once and stay in registers for the second
    // pass (the kernel was four to five dependent memory latencies long: 8.7 us for 20 KB; same per-thread summation order, so the same statistics)
    constexpr int KMAX = 12;
    if (total <= (long)KMAX * 256) {
        uint4 u[KMAX]; int px[KMAX], cc[KMAX];
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const long ii = (long)tid + 256 * k;
            px[k] = (int)(ii / cpc); cc[k] = (int)(ii % cpc);
            u[k] = ii < total ? *(const uint4*)(X + (long)px[k] * p.ldx + cc[k] * 8) : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            if ((long)tid + 256 * k < total) {
                float f[8];
                unpack8<T>(u[k], f);
#pragma unroll
                for (int e = 0; e < 8; ++e) { su += f[e]; sq = fmaf(f[e], f[e], sq); }
            }
        }
        su = wave_sum(su); sq = wave_sum(sq);
        if ((tid & 63) == 0) { red[0][tid >> 6] = su; red[1][tid >> 6] = sq; }
        __syncthreads();
        const float n = (float)p.HW * (float)cpg;
        const float mean = ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) / n;
        const float var = fmaxf(((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) / n - mean * mean, 0.f);
        const float rstd = rsqrtf(var + p.eps);
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            if ((long)tid + 256 * k < total) {
                float f[8];
                unpack8<T>(u[k], f);
                const float* gm = p.gamma + g * cpg + cc[k] * 8;
                const float* bt = p.beta + g * cpg + cc[k] * 8;
                const float4 g0 = *(const float4*)gm, g1 = *(const float4*)(gm + 4), b0 = *(const float4*)bt, b1 = *(const float4*)(bt + 4);
                const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float a = rstd * gg[e];
                    float y = fmaf(f[e], a, bb[e] - mean * a);
                    if (p.silu) y = silu_f(y);
                    f[e] = y;
                }
                *(uint4*)(Y + (long)px[k] * p.ldy + cc[k] * 8) = pack8<T>(f);
            }
        }
        return;
    }
    long i = tid;
    for (; i + 3 * 256 < total; i += 4 * 256) {        // four 16-byte loads in flight per thread
        uint4 u[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long ii = i + r * 256;
            const int pix = (int)(ii / cpc), ch = (int)(ii % cpc);
            u[r] = *(const uint4*)(X + (long)pix * p.ldx + ch * 8);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float f[8];
            unpack8<T>(u[r], f);
#pragma unroll
            for (int e = 0; e < 8; ++e) { su += f[e]; sq = fmaf(f[e], f[e], sq); }
        }
    }
    for (; i < total; i += 256) {
        const int pix = (int)(i / cpc), ch = (int)(i % cpc);
        float f[8];
        unpack8<T>(*(const uint4*)(X + (long)pix * p.ldx + ch * 8), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) { su += f[e]; sq = fmaf(f[e], f[e], sq); }
    }
    su = wave_sum(su); sq = wave_sum(sq);
    if ((tid & 63) == 0) { red[0][tid >> 6] = su; red[1][tid >> 6] = sq; }
    __syncthreads();
    const float n = (float)p.HW * (float)cpg;
    const float mean = ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) / n;
    const float var = fmaxf(((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) / n - mean * mean, 0.f);
    const float rstd = rsqrtf(var + p.eps);
    auto emit = [&](const uint4& u, const int pix, const int ch) __attribute__((always_inline)) {
        float f[8];
        unpack8<T>(u, f);
        const float* gm = p.gamma + g * cpg + ch * 8;
        const float* bt = p.beta + g * cpg + ch * 8;
        const float4 g0 = *(const float4*)gm, g1 = *(const float4*)(gm + 4), b0 = *(const float4*)bt, b1 = *(const float4*)(bt + 4);
        const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float a = rstd * gg[e];
            float y = fmaf(f[e], a, bb[e] - mean * a);
            if (p.silu) y = silu_f(y);
            f[e] = y;
        }
        *(uint4*)(Y + (long)pix * p.ldy + ch * 8) = pack8<T>(f);
    };
    i = tid;
    for (; i + 3 * 256 < total; i += 4 * 256) {
        uint4 u[4]; int px[4], cc[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long ii = i + r * 256;
            px[r] = (int)(ii / cpc); cc[r] = (int)(ii % cpc);
            u[r] = *(const uint4*)(X + (long)px[r] * p.ldx + cc[r] * 8);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) emit(u[r], px[r], cc[r]);
    }
    for (; i < total; i += 256) {
        const int pix = (int)(i / cpc), ch = (int)(i % cpc);
        emit(*(const uint4*)(X + (long)pix * p.ldx + ch * 8), pix, ch);
    }
}

// Producer-written statistics (GemmArgs::gn_partial) with more rows than gn_apply wants to fold per workgroup: [B][n_in][G][2] -> [B][GN_FOLD][G][2],
// each output row the fixed-order sum of a contiguous run of input rows (deterministic).  64 threads = (group, statistic) of G = 32.
__global__ __launch_bounds__(64) void gn_fold_kernel(const float* in, float* out, int n_in, int G) {
    const int f = blockIdx.x, b = blockIdx.y, t = threadIdx.x;
    if (t >= G * 2) return;
    const int per = (n_in + GN_FOLD - 1) / GN_FOLD;
    const int lo = f * per, hi = min(n_in, lo + per);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int c = lo;
    for (; c + 3 < hi; c += 4) {            // four loads in flight, summed in row order
        const float v0 = in[((long)b * n_in + c) * G * 2 + t], v1 = in[((long)b * n_in + c + 1) * G * 2 + t];
        const float v2 = in[((long)b * n_in + c + 2) * G * 2 + t], v3 = in[((long)b * n_in + c + 3) * G * 2 + t];
        a0 += v0; a1 += v1; a2 += v2; a3 += v3;
    }
    for (; c < hi; ++c) a0 += in[((long)b * n_in + c) * G * 2 + t];
    out[((long)b * GN_FOLD + f) * G * 2 + t] = (a0 + a1) + (a2 + a3);
}

bool gn_uses_small_kernel(int B, long HW, int C, int G) {
    static const long small_max = getenv("LDX_GN_SMALL_MAX") ? atol(getenv("LDX_GN_SMALL_MAX")) : 256 * 80;      // elements per (batch, group) the one-launch kernel takes
    return G > 0 && (C / G) % 8 == 0 && HW * (C / G) <= small_max && (long)G * B >= 32;
}

template <typename T>
static void launch_gn_t(const GroupNormArgs& a_in, hipStream_t s) {
    GroupNormArgs a = a_in;
    if (a.stats_chunks > 0) {           // statistics came with the producer's epilogue: apply only
        const GnGeom g = gn_geom(a.C);
        a.nchunk = a.stats_chunks;
        if (a.stats_chunks > GN_NCHUNK) {
            float* folded = a.partial + (size_t)a.B * a.stats_chunks * a.G * 2;
            hipLaunchKernelGGL(gn_fold_kernel, dim3(GN_FOLD, a.B), dim3(64), 0, s, a.partial, folded, a.stats_chunks, a.G);
            a.partial = folded; a.nchunk = GN_FOLD;
        }
        int nblk = (a.HW + g.RY * 8 - 1) / (g.RY * 8);
        if (nblk > 512) nblk = 512;
        if (nblk < 1) nblk = 1;
        dim3 grid2(nblk, a.B);
        switch (g.CH) {
            case 1: hipLaunchKernelGGL((gn_apply_kernel<T, 1>), grid2, dim3(256), 0, s, a, g.TX, g.RY, nblk); break;
            case 2: hipLaunchKernelGGL((gn_apply_kernel<T, 2>), grid2, dim3(256), 0, s, a, g.TX, g.RY, nblk); break;
            default: hipLaunchKernelGGL((gn_apply_kernel<T, 4>), grid2, dim3(256), 0, s, a, g.TX, g.RY, nblk); break;
        }
        return;
    }
    if (gn_uses_small_kernel(a.B, a.HW, a.C, a.G)) {
        hipLaunchKernelGGL((gn_small_kernel<T>), dim3(a.G, a.B), dim3(256), 0, s, a);
        return;
    }
    const GnGeom g = gn_geom(a.C);
    // enough pixel chunks to fill the chip (~1024 workgroups), but >= 4 pixels per thread row
    int nchunk = (1024 + a.B - 1) / a.B;
    const int maxc = (a.HW + g.RY * 4 - 1) / (g.RY * 4);
    if (nchunk > maxc) nchunk = maxc;
    if (nchunk > GN_NCHUNK) nchunk = GN_NCHUNK;
    if (nchunk < 1) nchunk = 1;
    a.nchunk = nchunk;
    dim3 grid1(nchunk, a.B);
    const size_t lds = (size_t)g.RY * a.C * 2 * sizeof(float);
    int nblk = (a.HW + g.RY * 8 - 1) / (g.RY * 8);
    if (nblk > 512) nblk = 512;
    if (nblk < 1) nblk = 1;
    dim3 grid2(nblk, a.B);
    switch (g.CH) {
        case 1:
            hipLaunchKernelGGL((gn_stats_kernel<T, 1>), grid1, dim3(256), lds, s, a, g.TX, g.RY);
            hipLaunchKernelGGL((gn_apply_kernel<T, 1>), grid2, dim3(256), 0, s, a, g.TX, g.RY, nblk);
            break;
        case 2:
            hipLaunchKernelGGL((gn_stats_kernel<T, 2>), grid1, dim3(256), lds, s, a, g.TX, g.RY);
            hipLaunchKernelGGL((gn_apply_kernel<T, 2>), grid2, dim3(256), 0, s, a, g.TX, g.RY, nblk);
            break;
        default:
            hipLaunchKernelGGL((gn_stats_kernel<T, 4>), grid1, dim3(256), lds, s, a, g.TX, g.RY);
            hipLaunchKernelGGL((gn_apply_kernel<T, 4>), grid2, dim3(256), 0, s, a, g.TX, g.RY, nblk);
            break;
    }
}

void launch_groupnorm(const GroupNormArgs& a, DType dt, hipStream_t s) {
    if (dt == DT_BF16) launch_gn_t<__bf16>(a, s); else launch_gn_t<_Float16>(a, s);
}

// ------------------------------------------------------------------------------------------
// LayerNorm: LPR lanes per row (16 / 32 / 64, so that a narrow row does not leave most of the wave idle and a
// wave keeps several rows' loads in flight), row held in registers (C <= 8 * LPR * NCH), two-pass mean / variance.
// Optional affine (gamma/beta) and optional per-batch adaLN modulation (Flux).
template <typename T, int LPR, int NCH>
__global__ __launch_bounds__(256) void ln_kernel(const LayerNormArgs p) {
    constexpr int RPW = 64 / LPR;                     // rows per wave
    const int lane = threadIdx.x & 63, sub = lane % LPR;
    const long row_raw = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + lane / LPR;
    const bool live = row_raw < p.rows;               // dead rows compute on the last row (all lanes stay active for the DPP reductions)
    const long row = live ? row_raw : p.rows - 1;
    const int nch = p.C >> 3;
    const T* __restrict__ x = (const T*)p.X + row * p.ldx;
    T* __restrict__ y = (T*)p.Y + row * p.ldy;
    float f[NCH][8];
    float sum = 0.f;
    // the row's chunks as UNCONDITIONAL loads issued together (a chunk beyond C reads chunk 0 and is ignored): behind `if (ch < nch)` hipcc emitted
    // load + s_waitcnt vmcnt(0) per chunk — NCH dependent HBM latencies per row (Flux C = 3072: six; 15.8 - 20.8 us for a 10.7 us transfer)
    uint4 raw[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) { const int ch = sub + LPR * i; raw[i] = *(const uint4*)(x + (ch < nch ? ch : 0) * 8); }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int ch = sub + LPR * i;
        unpack8<T>(raw[i], f[i]);
        if (ch < nch) {
#pragma unroll
            for (int e = 0; e < 8; ++e) sum += f[i][e];
        }
    }
    const float mean = p.rms ? 0.f : group_sum<LPR>(sum) / (float)p.C;
    float vs = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int ch = sub + LPR * i;
        if (ch < nch) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = f[i][e] - mean; vs = fmaf(d, d, vs); }
        }
    }
    const float rstd = rsqrtf(group_sum<LPR>(vs) / (float)p.C + p.eps);
    const long mb = p.scale ? (row / p.rows_per_batch) * (long)p.mod_ld : 0;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int ch = sub + LPR * i;
        if (ch < nch && live) {
            float o[8];
            // affine / modulation vectors as 16-byte loads (scalar per-element loads made this kernel VMEM-issue bound: 3x slower)
            auto ld8 = [&](const float* v, float (&d)[8]) {
                const float4 a = *(const float4*)(v + ch * 8), b = *(const float4*)(v + ch * 8 + 4);
                d[0] = a.x; d[1] = a.y; d[2] = a.z; d[3] = a.w; d[4] = b.x; d[5] = b.y; d[6] = b.z; d[7] = b.w;
            };
            float gg[8], bb[8], sc[8], sh[8];
            if (p.gamma) ld8(p.gamma, gg);
            if (p.gamma && p.beta) ld8(p.beta, bb);
            if (p.scale) { ld8(p.scale + mb, sc); ld8(p.shift + mb, sh); }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float v = (f[i][e] - mean) * rstd;
                if (p.gamma) v = v * gg[e] + (p.beta ? bb[e] : 0.f);
                if (p.scale) v = (1.0f + sc[e]) * v + sh[e];
                o[e] = v;
            }
            if (p.Y8) {          // MX fp8: a 32-element block = the chunks of 4 adjacent lanes (same i), C % 32 == 0 keeps quads uniform
                float amax = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) { o[e] = to_f32(from_f32<T>(o[e])); amax = fmaxf(amax, fabsf(o[e])); }
                amax = fmaxf(amax, dpp_f<0xB1>(amax));
                amax = fmaxf(amax, dpp_f<0x4E>(amax));
                const int ex = mx_scale_e8m0(amax);
                const float inv = mx_inv_scale(ex);
                *(uint2*)((char*)p.Y8 + row * p.ldy8 + ch * 8) = make_uint2(mx_pack4(o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv),
                                                                            mx_pack4(o[4] * inv, o[5] * inv, o[6] * inv, o[7] * inv));
                if ((sub & 3) == 0) { const int kb = ch >> 2; ((uint8_t*)p.S8)[((long)(kb >> 2) * p.s8_ld + row) * 4 + (kb & 3)] = (uint8_t)ex; }
            } else
            *(uint4*)(y + ch * 8) = pack8<T>(o);
        }
    }
}

template <typename T>
static void launch_ln_t(const LayerNormArgs& a, hipStream_t s) {
    const int nch = a.C >> 3;
    dim3 block(256);
#define LDX_LN(LPR, NCH) do { const int rpb = 4 * (64 / LPR); \
        hipLaunchKernelGGL((ln_kernel<T, LPR, NCH>), dim3((unsigned)((a.rows + rpb - 1) / rpb)), block, 0, s, a); } while (0)
    static const int force = getenv("LDX_LN_LPR") ? atoi(getenv("LDX_LN_LPR")) : 0;
    if (force == 64 && nch <= 192) { LDX_LN(64, 3); return; }
    if (force == 32 && nch <= 96) { LDX_LN(32, 3); return; }
    if (nch <= 48) LDX_LN(16, 3);              // C <= 384  (SD1.5 level 0: 320)
    else if (nch <= 96) LDX_LN(32, 3);         // C <= 768  (640, CLIP 768)
    else if (nch <= 192) LDX_LN(64, 3);        // C <= 1536 (1280)
    else if (nch <= 256) LDX_LN(64, 4);        // C <= 2048
    else if (nch <= 384) LDX_LN(64, 6);        // C <= 3072 (Flux)
    else LDX_LN(64, 8);                        // C <= 4096 (T5-XXL)
#undef LDX_LN
}

void launch_layernorm(const LayerNormArgs& a, DType dt, hipStream_t s) {
    if (a.rows <= 0) return;
    if (dt == DT_BF16) launch_ln_t<__bf16>(a, s); else launch_ln_t<_Float16>(a, s);
}

// ------------------------------------------------------------------------------------------
// Flux q/k: per-head RMSNorm + rotary embedding, in place.  One thread per 8 consecutive elements (4 rotary pairs, one
// 16-byte load / store); the D/8 threads of a head reduce the sum of squares with xor shuffles.
template <typename T>
__global__ __launch_bounds__(256) void qk_norm_rope_kernel(const QkRopeArgs p) {
    const int cpt = p.D >> 3;                                  // threads (16-B chunks) per head: 2..16, power of two
    const long total = (long)p.rows * 2 * p.H * cpt;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const bool live = idx < total;
    const long i = live ? idx : total - 1;
    const int ch = (int)(i % cpt);
    long r = i / cpt;
    const int h = (int)(r % p.H); r /= p.H;
    const int which = (int)(r & 1);                            // 0 = q, 1 = k
    const long row = r >> 1;
    T* __restrict__ ptr = (T*)p.QKV + row * p.ld + which * (p.H * p.D) + h * p.D + ch * 8;
    float x[8];
    unpack8<T>(*(const uint4*)ptr, x);
    float ss = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) ss = fmaf(x[e], x[e], ss);
    for (int o = cpt >> 1; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    const float rr = rsqrtf(ss / (float)p.D + p.eps);
    const float* sc = (which ? p.kscale : p.qscale) + ch * 8;
    const float4 s0 = *(const float4*)sc, s1 = *(const float4*)(sc + 4);
    const float sv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    const int tok = (int)(row % p.L);
    const long to = (long)tok * (p.D >> 1) + ch * 4;
    const float4 cs = *(const float4*)(p.cosT + to), sn = *(const float4*)(p.sinT + to);
    const float cv[4] = {cs.x, cs.y, cs.z, cs.w}, nv[4] = {sn.x, sn.y, sn.z, sn.w};
    float o[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float a = x[2 * q] * rr * sv[2 * q], b = x[2 * q + 1] * rr * sv[2 * q + 1];
        o[2 * q] = cv[q] * a - nv[q] * b;
        o[2 * q + 1] = nv[q] * a + cv[q] * b;
    }
    if (live) *(uint4*)ptr = pack8<T>(o);
}
void launch_qk_norm_rope(const QkRopeArgs& a, DType dt, hipStream_t s) {
    const long total = (long)a.rows * 2 * a.H * (a.D / 8);
    dim3 grid((unsigned)((total + 255) / 256)), block(256);
    if (dt == DT_BF16) hipLaunchKernelGGL((qk_norm_rope_kernel<__bf16>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((qk_norm_rope_kernel<_Float16>), grid, block, 0, s, a);
}

}  // namespace ldx
