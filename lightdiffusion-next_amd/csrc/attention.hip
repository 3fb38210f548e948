// Flash attention (online softmax) for gfx950 — replaces the reference's
// Attention.optimized_attention -> F.scaled_dot_product_attention call sites
// (Attention/AttentionMethods.py:107-150, used by CrossAttention.forward Attention.py:100-124;
// CLIP: clip/Clip.py:14-60 with a causal mask).
//
// Formulation (everything "transposed" so per-query statistics stay lane-local):
//   S^T[kv][q] = K[kv][:] . Q[q][:]          mfma(a = K frag, b = Q frag)
//   P          = exp2(S*c - m*c)             fp32 online softmax, c = scale*log2(e)
//   O^T[d][q] += V^T[d][kv] * P^T[kv][q]     mfma(a = V^T frag, b = P frag)
// With the 16x16x32 MFMA result layout (col = lane&15, rows = 4*(lane>>4)+r) the query index of
// every accumulator in both S^T and O^T is lane&15, so running max / rescale / denominator need
// no cross-lane traffic except one 4-lane max per 64-key block.  The k-slot <-> key permutation
// induced by feeding S^T accumulators straight back as the P operand is absorbed by the V operand:
// V stays row-major [key][d] in LDS (plain 16-byte staging writes) and each V^T fragment is two
// ds_read_b64_tr_b16 hardware-transposing reads (4 keys x 16 d per 16-lane group: keys 4g..4g+3 of
// two 16-key tiles).  Cross-lane max / sum use v_permlane16/32_swap (VALU), not LDS bpermute.
//
// The softmax denominator is not summed on the VALU: V carries a constant column of ones at d = D (the
// PV tile is padded to 16*DT > D anyway), so the same MFMA that accumulates O^T also accumulates
// l[q] = sum_k P[q][k] in row D of O^T, rescaled with O for free.
//
// Workgroup = 4 waves x 32 queries; K and V^T tiles of 64 keys double-buffered in LDS, global
// loads register-staged one block ahead.  Head dims D in {8..160}, D % 8 == 0: the QK^T
// contraction is padded to 32*KS, the PV output to 16*DT rows (zero-filled in LDS/registers).
#include <stdlib.h>
#include "ldx_device.h"
#include "ldx_kernels.h"

namespace ldx {

constexpr int AT_KV = 64;         // keys per block
// queries per wave = 16 * QT, per workgroup = 64 * QT (QT = 2, or 4 for small head dims where the
// accumulators fit: K/V staging and fragment reads are then amortised over twice the queries)

// LDS row strides.  K rows: KS*64 B + 32 B pad -> the four 16-lane groups of ds_read_b128 are conflict-free
// (brute-forced over the real lane groups).  V rows: >= DT*32 B and == 32 (mod 64) so the 8 key rows a
// 32-lane half touches in one ds_read_b64_tr_b16 tile the 64 banks.
template <int KS, int DT> struct AttnCfg {
    static constexpr int KROWB = KS * 64 + 32;
    static constexpr int VROWB = (DT * 32) % 64 == 0 ? DT * 32 + 32 : DT * 32;
};

typedef __attribute__((ext_vector_type(4))) short s16x4;
__device__ __forceinline__ uint2 lds_read_tr16(const char* p) {
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
    union { s16x4 v; uint2 u; } x; x.v = v; return x.u;
}
// value of lane^16 / lane^32 combined with own value; v_permlane*_swap with both operands = v leaves
// {own, partner} in the two results (order depends on the lane), so a commutative op needs no select.
__device__ __forceinline__ float quad_max(float v) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float quad_sum(float v) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

template <typename T, int KS, int DT, int QT>
__global__ __launch_bounds__(256, 2) void attn_kernel(const AttnArgs p) {
    constexpr int AT_QW = 16 * QT, AT_QB = 64 * QT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using V8 = typename Vec<T>::v8;
    constexpr int KROWB = AttnCfg<KS, DT>::KROWB;
    constexpr int VROWB = AttnCfg<KS, DT>::VROWB;
    constexpr int KBYTES = AT_KV * KROWB;
    constexpr int VBYTES = AT_KV * VROWB;
    constexpr int STAGE = KBYTES + VBYTES;
    constexpr int MAXCH = (KS * 4 > DT * 2) ? KS * 4 : DT * 2;     // upper bound on D/8
    constexpr int NLD = (AT_KV * MAXCH + 255) / 256;                // staging chunks per thread

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g4 = lane >> 4;
    // XCD-aware work order: block b runs on XCD b % 8 with a private 4 MiB L2, so each XCD is given a
    // contiguous run of (batch, head, q-block) work: its resident workgroups then stream the SAME head's
    // K/V (a few MB) instead of eight XCDs each thrashing over every head.
    const int nqb = (p.Nq + AT_QB - 1) / AT_QB;
    const int lin = xcd_remap(blockIdx.x, nqb * p.H * p.B);
    const int qblk = lin % nqb, hb = lin / nqb;
    const int h = hb % p.H, b = hb / p.H;
    const int q0 = qblk * AT_QB + wave * AT_QW;
    const int D = p.D, dch = D >> 3;           // 16-B chunks per head row
    const T* __restrict__ Qp = (const T*)p.Q + (long)b * p.Nq * p.ldq + h * D;
    const T* __restrict__ Kp = (const T*)p.K + (long)b * p.Mk * p.ldk + h * D;
    const T* __restrict__ Vp = (const T*)p.V + (long)b * p.Mk * p.ldv + h * D;
    T* __restrict__ Op = (T*)p.O + (long)b * p.Nq * p.ldo + h * D;
    const float c = p.scale * 1.44269504088896340736f;

    // zero both LDS stages once: K pad columns (d >= D) and V pad columns must read as 0 ...
    for (int i = tid; i < (2 * STAGE) / 16; i += 256) *(uint4*)(smem + i * 16) = make_uint4(0, 0, 0, 0);
    __syncthreads();
    // ... except V column D, which is the constant 1 that makes row D of O^T the softmax denominator.
    if (tid < 2 * AT_KV) *(T*)(smem + (tid >> 6) * STAGE + KBYTES + (tid & 63) * VROWB + D * 2) = (T)1.0f;

    // Q fragments (B operand): lane holds q = l15, d = ks*32 + g4*8 .. +7
    V8 qf[QT][KS];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const int q = q0 + qt * 16 + l15;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int ch = ks * 4 + g4;
            uint4 u = make_uint4(0, 0, 0, 0);
            if (q < p.Nq && ch < dch) u = *(const uint4*)(Qp + (long)q * p.ldq + ch * 8);
            asm volatile("" : "+v"(u.x), "+v"(u.y), "+v"(u.z), "+v"(u.w));   // wait for Q here, not at the first MFMA inside the loop (see attn32_kernel)
            qf[qt][ks] = as_v8<T>(u);
        }
    }

    f32x4 o[QT][DT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[qt][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float mrun[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) mrun[qt] = -INFINITY;

    // number of key blocks this workgroup needs (causal: keys <= last query of the block)
    int mk_eff = p.Mk;
    if (p.causal) mk_eff = min(p.Mk, qblk * AT_QB + AT_QB);
    const int nblk = (mk_eff + AT_KV - 1) / AT_KV;

    // K/V staging through buffer descriptors: per-lane byte offsets advance by one key block per iteration,
    // keys >= Mk (and staging slots beyond the 64 x D/8 tile) fall outside num_records and load as zeros —
    // no exec-mask branches or 64-bit address math in the loop.
    constexpr int OOB = (int)0x80000000;
    const __amdgpu_buffer_rsrc_t rK = __builtin_amdgcn_make_buffer_rsrc((void*)Kp, 0, (int)(((long)(p.Mk - 1) * p.ldk + D) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rV = __builtin_amdgcn_make_buffer_rsrc((void*)Vp, 0, (int)(((long)(p.Mk - 1) * p.ldv + D) * 2), 0x00020000);
    uint4 rk[NLD], rv[NLD];
    int ko[NLD], vo[NLD], lk[NLD], lv[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int idx = tid + i * 256;
        const int row = idx / dch, ch = idx - row * dch;
        const bool in_tile = row < AT_KV;
        ko[i] = in_tile ? (row * p.ldk + ch * 8) * 2 : OOB;
        vo[i] = in_tile ? (row * p.ldv + ch * 8) * 2 : OOB;
        lk[i] = in_tile ? row * KROWB + ch * 16 : -1;
        lv[i] = in_tile ? KBYTES + row * VROWB + ch * 16 : -1;
    }
    const int kstep = AT_KV * p.ldk * 2, vstep = AT_KV * p.ldv * 2;
    auto gload = [&](int blk) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const bool live = ko[i] != OOB;
            const auto a = __builtin_amdgcn_raw_buffer_load_b128(rK, live ? ko[i] + blk * kstep : OOB, 0, 0);
            const auto c2 = __builtin_amdgcn_raw_buffer_load_b128(rV, live ? vo[i] + blk * vstep : OOB, 0, 0);
            rk[i] = make_uint4(a[0], a[1], a[2], a[3]);
            rv[i] = make_uint4(c2[0], c2[1], c2[2], c2[3]);
        }
    };
    auto lstore = [&](int stage) {
        char* sB = smem + stage * STAGE;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            if (lk[i] >= 0) {
                *(uint4*)(sB + lk[i]) = rk[i];
                *(uint4*)(sB + lv[i]) = rv[i];      // V row-major; transposed on the read side
            }
        }
    };

    __syncthreads();      // zero / ones fill visible before the first tile lands
    if (nblk > 0) { gload(0); lstore(0); }
    __syncthreads();

    for (int blk = 0; blk < nblk; ++blk) {
        const int cur = blk & 1;
        const bool more = (blk + 1) < nblk;
        if (more) gload(blk + 1);
        const char* sK = smem + cur * STAGE;
        const char* sV = sK + KBYTES;
        const int kv0 = blk * AT_KV;

        // ---- S^T = K Q^T : 4 key tiles x 2 query tiles ----
        f32x4 s[QT][4];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
#pragma unroll
            for (int t = 0; t < 4; ++t) s[qt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const V8 kf = as_v8<T>(*(const uint4*)(sK + (t * 16 + l15) * KROWB + (ks * 4 + g4) * 16));
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) s[qt][t] = mfma16(kf, qf[qt][ks], s[qt][t]);
            }
        }
        __builtin_amdgcn_s_setprio(0);

        // ---- additive score bias (T5 relative positions): lane holds q = l15, keys 16t + 4*g4 + 0..3 ----
        if (p.bias) {
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                const int q = min(q0 + qt * 16 + l15, p.Nq - 1);
                const float* bp = p.bias + (long)h * p.bias_hs + (long)q * p.bias_ld + kv0 + g4 * 4;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float4 bv = *(const float4*)(bp + t * 16);
                    s[qt][t][0] += bv.x; s[qt][t][1] += bv.y; s[qt][t][2] += bv.z; s[qt][t][3] += bv.w;
                }
            }
        }

        // ---- masking (block-uniform fast path) ----
        const bool need_mask = (kv0 + AT_KV > p.Mk) || (p.causal && (kv0 + AT_KV - 1 > qblk * AT_QB));
        if (need_mask) {
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                const int q = q0 + qt * 16 + l15;
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int kv = kv0 + t * 16 + g4 * 4 + r;
                        const bool dead = kv >= p.Mk || (p.causal && kv > q);
                        if (dead) s[qt][t][r] = -INFINITY;
                    }
            }
        }

        // ---- online softmax; P packed as B-operand fragments ----
        V8 pf[QT][2];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            float mx = s[qt][0][0];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[qt][t][r]);
            mx = quad_max(mx);
            const float mnew = fmaxf(mrun[qt], mx);
            // rows with every key masked so far keep mnew = -inf; guard the (-inf) - (-inf) case
            const float mc = (mnew == -INFINITY) ? 0.f : mnew * c;
            const float alpha = __builtin_amdgcn_exp2f(mrun[qt] * c - mc);
            mrun[qt] = mnew;
            float pv[4][4];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = __builtin_amdgcn_exp2f(fmaf(s[qt][t][r], c, -mc));
                    pv[t][r] = e;
                }
            // the running maximum settles after the first few key blocks: skip the rescale of O^T (DT x 4 multiplies per lane)
            // whenever no query of this wave moved its maximum (alpha == 1 everywhere) — exact, wave-uniform branch
            if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    o[qt][dt][0] *= alpha; o[qt][dt][1] *= alpha; o[qt][dt][2] *= alpha; o[qt][dt][3] *= alpha;
                }
            }
#pragma unroll
            for (int ks2 = 0; ks2 < 2; ++ks2) {
                V8 f;
#pragma unroll
                for (int r = 0; r < 4; ++r) { f[r] = (T)pv[2 * ks2][r]; f[4 + r] = (T)pv[2 * ks2 + 1][r]; }
                pf[qt][ks2] = f;
            }
        }

        // ---- O^T += V^T P^T ----
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
            for (int ks2 = 0; ks2 < 2; ++ks2) {
                // 16-lane group g4 fetches keys ks2*32 + 4*g4 + {0..3} (+16 for the second read), d = dt*16 + 0..15:
                // lane i of the group addresses 4 contiguous d of key (i>>2) and receives column d = dt*16 + i.
                const char* vp = sV + (ks2 * 32 + g4 * 4 + (l15 >> 2)) * VROWB + (dt * 16 + (l15 & 3) * 4) * 2;
                U128 vf;
                vf.d[0] = lds_read_tr16(vp);
                vf.d[1] = lds_read_tr16(vp + 16 * VROWB);
                const V8 v8 = as_v8<T>(vf.u);
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) o[qt][dt] = mfma16(v8, pf[qt][ks2], o[qt][dt]);
            }
        }
        __builtin_amdgcn_s_setprio(0);

        if (more) lstore(cur ^ 1);
        __syncthreads();
    }

    // ---- finalize: O = O^T / l ; lane holds q = l15, d = dt*16 + 4*g4 + r ----
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        // denominator = O^T row D: tile DT-1, lanes with g4 == (D % 16) / 4, register 0
        const float l = quad_sum(g4 == ((D & 15) >> 2) ? o[qt][DT - 1][0] : 0.f);
        const float inv = (l > 0.f) ? 1.0f / l : 0.f;
        const int q = q0 + qt * 16 + l15;
        if (q >= p.Nq) continue;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const int d = dt * 16 + g4 * 4;
            if (d < D)
                *(uint2*)(Op + (long)q * p.ldo + d) =
                    pack4<T>(o[qt][dt][0] * inv, o[qt][dt][1] * inv, o[qt][dt][2] * inv, o[qt][dt][3] * inv);
        }
    }
}


// Maximum of the 32 scores a lane holds for one query (two 32x32 S^T tiles): four independent v_max3 chains of depth 4 instead of one
// serial chain of depth 16 — a wave issues in order, so the serial form exposed ~16 dependent-VALU latencies per query tile and key block.
__device__ __forceinline__ float max32_tree(const float (&a)[16], const float (&b)[16]) {
    float m0 = fmaxf(a[0], a[1]), m1 = fmaxf(a[8], a[9]), m2 = fmaxf(b[0], b[1]), m3 = fmaxf(b[8], b[9]);
#pragma unroll
    for (int j = 1; j < 4; ++j) {
        m0 = fmaxf(fmaxf(m0, a[2 * j]), a[2 * j + 1]);
        m1 = fmaxf(fmaxf(m1, a[8 + 2 * j]), a[8 + 2 * j + 1]);
        m2 = fmaxf(fmaxf(m2, b[2 * j]), b[2 * j + 1]);
        m3 = fmaxf(fmaxf(m3, b[8 + 2 * j]), b[8 + 2 * j + 1]);
    }
    return fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
}

__device__ __forceinline__ float max32_tree_v(const f32x16& a, const f32x16& b) {       // same tree on the accumulator vectors themselves
    float m0 = fmaxf(a[0], a[1]), m1 = fmaxf(a[8], a[9]), m2 = fmaxf(b[0], b[1]), m3 = fmaxf(b[8], b[9]);
#pragma unroll
    for (int j = 1; j < 4; ++j) {
        m0 = fmaxf(fmaxf(m0, a[2 * j]), a[2 * j + 1]);
        m1 = fmaxf(fmaxf(m1, a[8 + 2 * j]), a[8 + 2 * j + 1]);
        m2 = fmaxf(fmaxf(m2, b[2 * j]), b[2 * j + 1]);
        m3 = fmaxf(fmaxf(m3, b[8 + 2 * j]), b[8 + 2 * j + 1]);
    }
    return fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
}

// ------------------------------------------------------------------------------------------------------------
// 32x32x16 variant for 32 < D <= 48 (SD1.5 level 0: D = 40, the largest kernel of a step).  On gfx950 the 16x16x32 MFMA
// issues every ~21.5 cycles (75 % of peak) while 32x32x16 issues every 32.4 cycles for twice the work (profiles/ubench/
// mfma_rate.hip), and at D = 40 the 16x16x32 QK^T pads the contraction to 64.  Here QK^T contracts over 48 (3 k-steps of 16):
// 12 MFMAs per 64 keys x 64 queries instead of 32, PV 16 instead of 24 — 907 instead of 1204 issue cycles per key block.
//
// Same transposed formulation: S^T[key][q] and O^T[d][q], so lane l owns query l & 31 of a 32-query tile (both halves of the
// wave hold the same queries, different keys / d rows) and the softmax state is lane-local up to one permlane32 exchange.
// C layout of a 32x32 tile: lane l holds column l & 31, rows 8*(r>>2) + 4*(l>>5) + (r&3), r = 0..15.  Feeding S^T registers
// 8*(s&1)..+7 of key tile s>>1 straight back as the P^T operand of k-step s makes k-slot 8h+e stand for key
// 16s + (e < 4 ? 4h + e : 8 + 4h + e - 4); the V^T operand is gathered in exactly that order by two ds_read_b64_tr_b16 per
// 16-lane group (keys 16s + 4h + 0..3 and 16s + 8 + 4h + 0..3, 16 d columns each).
template <typename T, int KVB, int KROWB, int VROWB, int VAR = 0>       // KVB keys per LDS stage (64 or 128): one barrier + one staging round per KVB keys
__global__ __launch_bounds__(256, 2) void attn32_kernel(const AttnArgs p) {
    constexpr int QW = 64, QB = 256;                 // queries per wave / workgroup
    // LDS row strides (PMC: SQ_LDS_BANK_CONFLICT).  K rows are read 16 B per lane by 16 different rows at one chunk column: the
    // stride must be an ODD number of 16-B chunks (9: 144 B).  V rows are read by ds_read_b64_tr_b16, 8 lanes x 8 B per row and
    // 4 rows per 32-lane group: the four 64-B windows tile the 64 banks when the stride is 64 B times an odd number (3: 192 B).
    constexpr int KBYTES = KVB * KROWB, VBYTES = KVB * VROWB, STAGE = KBYTES + VBYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using V8 = typename Vec<T>::v8;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h2 = lane >> 5, l15 = lane & 15, g16 = lane >> 4;
    const int nqb = (p.Nq + QB - 1) / QB;
    const int lin = xcd_remap(blockIdx.x, nqb * p.H * p.B);
    const int qblk = lin % nqb, hb = lin / nqb;
    const int h = hb % p.H, b = hb / p.H;
    const int q0 = qblk * QB + wave * QW;
    const int D = p.D, dch = D >> 3;
    const T* __restrict__ Qp = (const T*)p.Q + (long)b * p.Nq * p.ldq + h * D;
    const T* __restrict__ Kp = (const T*)p.K + (long)b * p.Mk * p.ldk + h * D;
    const T* __restrict__ Vp = (const T*)p.V + (long)b * p.Mk * p.ldv + h * D;
    T* __restrict__ Op = (T*)p.O + (long)b * p.Nq * p.ldo + h * D;
    const float c = p.scale * 1.44269504088896340736f;

    for (int i = tid; i < (2 * STAGE) / 16; i += 256) *(uint4*)(smem + i * 16) = make_uint4(0, 0, 0, 0);
    __syncthreads();
    for (int i = tid; i < 2 * KVB; i += 256) *(T*)(smem + (i / KVB) * STAGE + KBYTES + (i % KVB) * VROWB + D * 2) = (T)1.0f;       // ones column of V at d = D

    // Q fragments (B operand): lane holds q = l31, d = 16*ks + 8*h2 .. +7
    V8 qf[2][3];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int q = q0 + qt * 32 + l31;
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            const int ch = 2 * ks + h2;
            uint4 u = make_uint4(0, 0, 0, 0);
            if (q < p.Nq && ch < dch) u = *(const uint4*)(Qp + (long)q * p.ldq + ch * 8);
            asm volatile("" : "+v"(u.x), "+v"(u.y), "+v"(u.z), "+v"(u.w));   // consume the load HERE: otherwise hipcc's waitcnt pass, which sees the Q loads still pending on the
                                                     // nblk == 0 / exec-masked paths around the first lstore, puts the wait at the first QK^T MFMA INSIDE the
                                                     // loop as vmcnt(0) — right behind the next block's staging loads, whose latency then stalls every iteration
            qf[qt][ks] = as_v8<T>(u);
        }
    }
    f32x16 o[2][2];                                  // [q tile][d tile of 32]
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    o[0][0] = zero16; o[0][1] = zero16; o[1][0] = zero16; o[1][1] = zero16;
    float mrun[2] = {-INFINITY, -INFINITY};
    const int nblk = (p.Mk + KVB - 1) / KVB;

    constexpr int OOB = (int)0x80000000;
    const __amdgpu_buffer_rsrc_t rK = __builtin_amdgcn_make_buffer_rsrc((void*)Kp, 0, (int)(((long)(p.Mk - 1) * p.ldk + D) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rV = __builtin_amdgcn_make_buffer_rsrc((void*)Vp, 0, (int)(((long)(p.Mk - 1) * p.ldv + D) * 2), 0x00020000);
    constexpr int NLD = (KVB * 6 + 255) / 256;       // up to 6 chunks per row (D <= 48)
    uint4 rk[NLD], rv[NLD];
    int ko[NLD], vo[NLD], lk[NLD], lv[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int idx = tid + i * 256;
        const int row = idx / dch, ch = idx - row * dch;
        const bool in_tile = row < KVB;
        ko[i] = in_tile ? (row * p.ldk + ch * 8) * 2 : OOB;
        vo[i] = in_tile ? (row * p.ldv + ch * 8) * 2 : OOB;
        lk[i] = in_tile ? row * KROWB + ch * 16 : -1;
        lv[i] = in_tile ? KBYTES + row * VROWB + ch * 16 : -1;
    }
    const int kstep = KVB * p.ldk * 2, vstep = KVB * p.ldv * 2;
    auto gload = [&](int blk) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const bool live = ko[i] != OOB;
            const auto a = __builtin_amdgcn_raw_buffer_load_b128(rK, live ? ko[i] + blk * kstep : OOB, 0, 0);
            const auto c2 = __builtin_amdgcn_raw_buffer_load_b128(rV, live ? vo[i] + blk * vstep : OOB, 0, 0);
            rk[i] = make_uint4(a[0], a[1], a[2], a[3]);
            rv[i] = make_uint4(c2[0], c2[1], c2[2], c2[3]);
        }
    };
    auto lstore = [&](int stage) {
        char* sB = smem + stage * STAGE;
#pragma unroll
        for (int i = 0; i < NLD; ++i)
            if (lk[i] >= 0) { *(uint4*)(sB + lk[i]) = rk[i]; *(uint4*)(sB + lv[i]) = rv[i]; }
    };

    __syncthreads();
    if (nblk > 0) { gload(0); lstore(0); }
    __syncthreads();

    for (int blk = 0; blk < nblk; ++blk) {
        const int cur = blk & 1;
        const bool more = (blk + 1) < nblk;
        if (more) gload(blk + 1);
        auto process = [&](const int sub) __attribute__((always_inline)) {
        const char* sK = smem + cur * STAGE + sub * 64 * KROWB;
        const char* sV = smem + cur * STAGE + KBYTES + sub * 64 * VROWB;
        const int kv0 = blk * KVB + sub * 64;

        // ---- S^T = K Q^T : 2 key tiles x 2 query tiles x 3 k-steps ----
        f32x16 s[2][2];                              // [key tile][q tile]
        s[0][0] = zero16; s[0][1] = zero16; s[1][0] = zero16; s[1][1] = zero16;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 3; ++ks)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                const V8 kf = as_v8<T>(*(const uint4*)(sK + (kt * 32 + l31) * KROWB + (2 * ks + h2) * 16));
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) s[kt][qt] = mfma32(kf, qf[qt][ks], s[kt][qt]);
            }
        __builtin_amdgcn_s_setprio(0);

        // scores as plain scalars from here on (element writes into a 16-wide vector made hipcc route it through scratch)
        float sv[2][2][16];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                for (int r = 0; r < 16; ++r) sv[kt][qt][r] = s[kt][qt][r];
        if (kv0 + 64 > p.Mk) {                       // ragged last key block (wave-uniform branch)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kv = kv0 + kt * 32 + 8 * (r >> 2) + 4 * h2 + (r & 3);
                    if (kv >= p.Mk) { sv[kt][0][r] = -INFINITY; sv[kt][1][r] = -INFINITY; }
                }
            asm volatile("" ::: "memory");           // keep this a branch: if-conversion would run 64 selects every key block
        }

        // VAR 1: the V^T fragments of this key block are fetched BEFORE the softmax, so that their LDS latency hides under its VALU work
        // instead of stalling the first PV MFMAs (+32 VGPRs)
        V8 vpre[VAR == 1 ? 4 : 1][VAR == 1 ? 2 : 1];
        if constexpr (VAR == 1) {
#pragma unroll
            for (int st = 0; st < 4; ++st)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const char* vp = sV + (16 * st + 4 * (g16 >> 1) + (l15 >> 2)) * VROWB + (dt * 32 + 16 * (g16 & 1) + (l15 & 3) * 4) * 2;
                    U128 vf;
                    vf.d[0] = lds_read_tr16(vp);
                    vf.d[1] = lds_read_tr16(vp + 8 * VROWB);
                    vpre[st][dt] = as_v8<T>(vf.u);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- online softmax; P packed as B-operand fragments: k-step st takes registers 8*(st&1)..+7 of key tile st>>1 ----
        V8 pf[2][4];
        float mcs[2], alphas[2];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {             // statistics of BOTH query tiles first (independent chains), one rescale branch for the pair
            float mx = max32_tree(sv[0][qt], sv[1][qt]);
            {   // the other half of the wave holds the other 32 keys of the same query
                auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
                mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
            }
            const float mnew = fmaxf(mrun[qt], mx);
            mcs[qt] = (mnew == -INFINITY) ? 0.f : mnew * c;
            alphas[qt] = __builtin_amdgcn_exp2f(mrun[qt] * c - mcs[qt]);
            mrun[qt] = mnew;
        }
        if (__builtin_amdgcn_ballot_w64(alphas[0] != 1.0f || alphas[1] != 1.0f) != 0) {
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                o[qt][0] = o[qt][0] * alphas[qt];    // whole-vector scale: no element inserts
                o[qt][1] = o[qt][1] * alphas[qt];
            }
        }
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            const float mc = mcs[qt];
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                V8 f;
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = (T)__builtin_amdgcn_exp2f(fmaf(sv[st >> 1][qt][8 * (st & 1) + e], c, -mc));
                pf[qt][st] = f;
            }
        }

        // ---- O^T += V^T P^T : per k-step and d tile one V^T fragment (two transposing reads), used by both query tiles ----
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                // 16-lane group g16: h' = g16 >> 1, d columns dt*32 + 16*(g16 & 1) + 0..15; lane i of the group addresses 4 contiguous d
                // of key (i >> 2) and receives column d = .. + i for keys K0 + 0..3
                const char* vp = sV + (16 * st + 4 * (g16 >> 1) + (l15 >> 2)) * VROWB + (dt * 32 + 16 * (g16 & 1) + (l15 & 3) * 4) * 2;
                V8 v8;
                if constexpr (VAR == 1) v8 = vpre[st][dt];
                else {
                    U128 vf;
                    vf.d[0] = lds_read_tr16(vp);
                    vf.d[1] = lds_read_tr16(vp + 8 * VROWB);
                    v8 = as_v8<T>(vf.u);
                }
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) o[qt][dt] = mfma32(v8, pf[qt][st], o[qt][dt]);
            }
        __builtin_amdgcn_s_setprio(0);
        };
        process(0);
        if (KVB == 128 && blk * KVB + 64 < p.Mk) process(1);

        if (more) lstore(cur ^ 1);
        __syncthreads();
    }

    // ---- finalize: O = O^T / l ; denominator = O^T row D (the ones column of V) ----
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        // row D lives in d tile D / 32, at r with 8*(r>>2) + 4*h + (r&3) == D % 32 on the half h = ((D % 32) >> 2) & 1
        const int dd = D & 31;
        // 32 < D <= 48, D % 4 == 0: d tile 1, register 4*(dd>>3); selected with static indices (a runtime index would put o[] in scratch)
        const float cand = dd < 8 ? o[qt][1][0] : (dd < 16 ? o[qt][1][4] : o[qt][1][8]);
        const float mine = (h2 == ((dd >> 2) & 1)) ? cand : 0.f;
        auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mine), __float_as_uint(mine), false, false);
        const float l = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
        const float inv = (l > 0.f) ? 1.0f / l : 0.f;
        const int q = q0 + qt * 32 + l31;
        if (q >= p.Nq) continue;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int d = dt * 32 + 8 * rq + 4 * h2;
                if (d < D)
                    *(uint2*)(Op + (long)q * p.ldo + d) = pack4<T>(o[qt][dt][4 * rq] * inv, o[qt][dt][4 * rq + 1] * inv, o[qt][dt][4 * rq + 2] * inv, o[qt][dt][4 * rq + 3] * inv);
            }
    }
}

template <typename T, int KVB, int KROWB, int VROWB, int VAR = 0>
static void launch_attn32_k(const AttnArgs& a, hipStream_t s) {
    const size_t lds = 2 * KVB * (KROWB + VROWB);
    static DevOnce once;
    set_dyn_lds(once, (const void*)attn32_kernel<T, KVB, KROWB, VROWB, VAR>, (int)lds);
    dim3 grid(((a.Nq + 255) / 256) * a.H * a.B);
    hipLaunchKernelGGL((attn32_kernel<T, KVB, KROWB, VROWB, VAR>), grid, dim3(256), lds, s, a);
}
#include "attn32ap.inc"
template <typename T>
static void launch_attn32(const AttnArgs& a, hipStream_t s) {
    // 8-wave two-group kernel (round 3, attn32ap.inc) when its 512-query workgroups still fill the chip.  LDX_ATTN32_AP: 0 off, 1 (default) phases
    // separated by the barriers only (hipcc interleaves the next PV MFMAs with the softmax), 2 strict phases, 3 strict + s_setprio around the MFMAs.
    // Same box, B2 H8 N16384 D40: attn32_kernel 1.197 ms, strict 1.14-1.21, default 1.12-1.15; step 60.02 -> 61.18 it/s.
    static const int ap = getenv("LDX_ATTN32_AP") ? atoi(getenv("LDX_ATTN32_AP")) : 1;
    static const long ap_minwg = getenv("LDX_ATTN32_AP_MINWG") ? atol(getenv("LDX_ATTN32_AP_MINWG")) : 256;
    if (ap && a.Mk >= 64 && (long)((a.Nq + 511) / 512) * a.H * a.B >= ap_minwg) {
        switch (ap) {
            case 2: launch_attn32ap<T, 0>(a, s); break;
            case 3: launch_attn32ap<T, 1>(a, s); break;
#ifdef LDX_ATTN_ABLATE                                    // measured-and-not-adopted variants (6, 7: correct, bit-identical) and timing ablations
            case 6: launch_attn32ap<T, 66>(a, s); break;      // (wrong results) of profiles/ubench/README.md round 3.  6: split softmax (VAR & 64), +6 %
            case 7: launch_attn32ap<T, 130>(a, s); break;     // 7: K fragments read a phase early (VAR & 128), +-0.5 %
            case 4: launch_attn32ap<T, 4>(a, s); break;       // no MFMAs
            case 8: launch_attn32ap<T, 8>(a, s); break;       // no softmax
            case 12: launch_attn32ap<T, 12>(a, s); break;     // neither: staging + barriers + fragment reads
            case 34: launch_attn32ap<T, 34>(a, s); break;     // default schedule without the 64 fma of exp2(s * c - m * c)
#endif
            default: launch_attn32ap<T, 2>(a, s); break;
        }
        return;
    }
    static const int kvb = getenv("LDX_ATTN32_KVB") ? atoi(getenv("LDX_ATTN32_KVB")) : 64;       // experiment switch
    // strides 160 / 160 had 2-way conflicts on every fragment read (SQ_LDS_BANK_CONFLICT 88.1 M -> 21.0 M cycles per launch with 144 / 192;
    // same time at D = 40, which is VALU / MFMA bound)
    static const int var = getenv("LDX_ATTN32_VAR") ? atoi(getenv("LDX_ATTN32_VAR")) : 1;               // 1: V^T fragments prefetched before the softmax (1.219 -> 1.206 ms at B2 H8 N16384, twice on one box); 0: read at the PV MFMAs
    if (var == 1) { launch_attn32_k<T, 64, 144, 192, 1>(a, s); return; }
    if (kvb == 128 && a.Mk >= 1024) launch_attn32_k<T, 128, 144, 192>(a, s); else launch_attn32_k<T, 64, 144, 192>(a, s);
}


// Generalisation of attn32_kernel to the other head dims of the path (D = 64 CLIP/T5-sized, 80 and 160 for SD1.5 levels 1-3,
// 128 for Flux): NKS k-steps of 16 over D (<= 16*NKS), NDT d-tiles of 32 holding D + 1 rows (the +1 is the ones row), QT query
// tiles of 32 per wave.  Same register recycling S^T -> P^T and the same V^T gather; K rows are padded to an odd multiple of 32 B
// and V rows to 64*NDT + 32 B so that both fragment reads stay bank-conflict free.
// ONES: denominator from a ones column of V (row D of O^T, needs D + 1 <= 32*NDT); otherwise (D a multiple of 32: the ones row would
// cost a whole extra d tile) the fp32 P values are summed on the VALU per lane and the two halves are added once at the end.
// LDS row strides of the generic kernel, same rules as attn32_kernel: K rows an odd number of 16-B chunks, V rows 64 B times an
// odd number (LDX_G32_OLD_ROWS at build time restores the first layout for A/B)
#ifdef LDX_G32_OLD_ROWS
#define G32_KROW(NKS) (((NKS) & 1) ? (NKS) * 32 : (NKS) * 32 + 32)
#define G32_VROW(NDT) ((NDT) * 64 + 32)
#else
#define G32_KROW(NKS) ((NKS) * 32 + 16)
#define G32_VROW(NDT) (((NDT) & 1) ? (NDT) * 64 : (NDT) * 64 + 64)
#endif
template <typename T, int NKS, int NDT, int QT, bool ONES, bool KPFON = true>
__global__ __launch_bounds__(256, 2) void attn32g_kernel(const AttnArgs p) {
    constexpr int QW = 32 * QT, QB = 4 * QW;
    constexpr int KROW = G32_KROW(NKS), VROW = G32_VROW(NDT);
    constexpr int KBYTES = AT_KV * KROW, VBYTES = AT_KV * VROW, STAGE = KBYTES + VBYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using V8 = typename Vec<T>::v8;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h2 = lane >> 5, l15 = lane & 15, g16 = lane >> 4;
    const int nqb = (p.Nq + QB - 1) / QB;
    const int lin = xcd_remap(blockIdx.x, nqb * p.H * p.B);
    const int qblk = lin % nqb, hb = lin / nqb;
    const int h = hb % p.H, b = hb / p.H;
    const int q0 = qblk * QB + wave * QW;
    const int D = p.D, dch = D >> 3;
    const T* __restrict__ Qp = (const T*)p.Q + (long)b * p.Nq * p.ldq + h * D;
    const T* __restrict__ Kp = (const T*)p.K + (long)b * p.Mk * p.ldk + h * D;
    const T* __restrict__ Vp = (const T*)p.V + (long)b * p.Mk * p.ldv + h * D;
    T* __restrict__ Op = (T*)p.O + (long)b * p.Nq * p.ldo + h * D;
    const float c = p.scale * 1.44269504088896340736f;

    for (int i = tid; i < (2 * STAGE) / 16; i += 256) *(uint4*)(smem + i * 16) = make_uint4(0, 0, 0, 0);
    __syncthreads();
    if (ONES && tid < 2 * AT_KV) *(T*)(smem + (tid >> 6) * STAGE + KBYTES + (tid & 63) * VROW + D * 2) = (T)1.0f;       // ones column of V at d = D

    V8 qf[QT][NKS];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const int q = q0 + qt * 32 + l31;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const int ch = 2 * ks + h2;
            uint4 u = make_uint4(0, 0, 0, 0);
            if (q < p.Nq && ch < dch) u = *(const uint4*)(Qp + (long)q * p.ldq + ch * 8);
            asm volatile("" : "+v"(u.x), "+v"(u.y), "+v"(u.z), "+v"(u.w));   // wait for Q here, not at the first MFMA inside the loop (see attn32_kernel)
            qf[qt][ks] = as_v8<T>(u);
        }
    }
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 o[QT][NDT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) o[qt][dt] = zero16;
    float mrun[QT], lsum[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) { mrun[qt] = -INFINITY; lsum[qt] = 0.f; }
    const int nblk = (p.Mk + AT_KV - 1) / AT_KV;

    constexpr int OOB = (int)0x80000000;
    const __amdgpu_buffer_rsrc_t rK = __builtin_amdgcn_make_buffer_rsrc((void*)Kp, 0, (int)(((long)(p.Mk - 1) * p.ldk + D) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rV = __builtin_amdgcn_make_buffer_rsrc((void*)Vp, 0, (int)(((long)(p.Mk - 1) * p.ldv + D) * 2), 0x00020000);
    constexpr int NLD = (AT_KV * NKS * 2 + 255) / 256;       // up to 2*NKS chunks per row
    uint4 rk[NLD], rv[NLD];
    int ko[NLD], vo[NLD], lk[NLD], lv[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int idx = tid + i * 256;
        const int row = idx / dch, ch = idx - row * dch;
        const bool in_tile = row < AT_KV;
        ko[i] = in_tile ? (row * p.ldk + ch * 8) * 2 : OOB;
        vo[i] = in_tile ? (row * p.ldv + ch * 8) * 2 : OOB;
        lk[i] = in_tile ? row * KROW + ch * 16 : -1;
        lv[i] = in_tile ? KBYTES + row * VROW + ch * 16 : -1;
    }
    const int kstep = AT_KV * p.ldk * 2, vstep = AT_KV * p.ldv * 2;
    auto gload = [&](int blk) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const bool live = ko[i] != OOB;
            const auto a = __builtin_amdgcn_raw_buffer_load_b128(rK, live ? ko[i] + blk * kstep : OOB, 0, 0);
            const auto c2 = __builtin_amdgcn_raw_buffer_load_b128(rV, live ? vo[i] + blk * vstep : OOB, 0, 0);
            rk[i] = make_uint4(a[0], a[1], a[2], a[3]);
            rv[i] = make_uint4(c2[0], c2[1], c2[2], c2[3]);
        }
    };
    auto lstore = [&](int stage) {
        char* sB = smem + stage * STAGE;
#pragma unroll
        for (int i = 0; i < NLD; ++i)
            if (lk[i] >= 0) { *(uint4*)(sB + lk[i]) = rk[i]; *(uint4*)(sB + lv[i]) = rv[i]; }
    };

    __syncthreads();
#ifndef LDX_ATTN_NO_T14
    // staging schedule as in gemm.hip (guide T14): the registers hold the NEXT key block; it is written to the other stage at the top
    // of an iteration, right after the barrier, and the loads of the block after it are re-issued before the compute phase.
    // Same-box A/B: D = 128 0.315 -> 0.303 ms (Flux), D = 80 / 160 unchanged; the D = 40 kernel lost 1 % with it and keeps
    // load -> compute -> write -> barrier
    if (nblk > 0) { gload(0); lstore(0); if (nblk > 1) gload(1); }
#else
    if (nblk > 0) { gload(0); lstore(0); }
#endif
    __syncthreads();

    for (int blk = 0; blk < nblk; ++blk) {
        const int cur = blk & 1;
        const bool more = (blk + 1) < nblk;
#ifndef LDX_ATTN_NO_T14
        if (more) { lstore(cur ^ 1); if (blk + 2 < nblk) gload(blk + 2); }
#else
        if (more) gload(blk + 1);
#endif
        const char* sK = smem + cur * STAGE;
        const char* sV = sK + KBYTES;
        const int kv0 = blk * AT_KV;

        f32x16 s[2][QT];                             // [key tile][q tile]
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) s[kt][qt] = zero16;
        // K fragments run KPF k-steps ahead of the MFMAs that consume them (a ring of KPF x 2 fragments).  Left to itself hipcc sinks every
        // ds_read_b128 to its first use: read x2 / wait / MFMA / read x2 / wait ... — with one MFMA per fragment (QT = 1) each QK^T MFMA then waits
        // out most of an LDS round trip (D = 128: 16 MFMAs, 8 exposed waits per key block).
        constexpr int KPF = (NKS >= 10 ? 2 : (NKS < 4 ? NKS : 4)) * (KPFON ? 1 : 0);
        __builtin_amdgcn_s_setprio(1);
        if constexpr (KPF > 0) {
            V8 kfr[KPF][2];
#pragma unroll
            for (int d = 0; d < KPF; ++d)
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) kfr[d][kt] = as_v8<T>(*(const uint4*)(sK + (kt * 32 + l31) * KROW + (2 * d + h2) * 16));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt) s[kt][qt] = mfma32(kfr[ks % KPF][kt], qf[qt][ks], s[kt][qt]);
                if (ks + KPF < NKS) {
#pragma unroll
                    for (int kt = 0; kt < 2; ++kt) kfr[ks % KPF][kt] = as_v8<T>(*(const uint4*)(sK + (kt * 32 + l31) * KROW + (2 * (ks + KPF) + h2) * 16));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                const V8 kf = as_v8<T>(*(const uint4*)(sK + (kt * 32 + l31) * KROW + (2 * ks + h2) * 16));
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) s[kt][qt] = mfma32(kf, qf[qt][ks], s[kt][qt]);
            }
        }
        __builtin_amdgcn_s_setprio(0);

        float sv[2][QT][16];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int qt = 0; qt < QT; ++qt)
#pragma unroll
                for (int r = 0; r < 16; ++r) sv[kt][qt][r] = s[kt][qt][r];
        if (kv0 + AT_KV > p.Mk) {                    // ragged last key block (wave-uniform branch)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kv = kv0 + kt * 32 + 8 * (r >> 2) + 4 * h2 + (r & 3);
                    if (kv >= p.Mk) {
#pragma unroll
                        for (int qt = 0; qt < QT; ++qt) sv[kt][qt][r] = -INFINITY;
                    }
                }
            asm volatile("" ::: "memory");
        }

        V8 pf[QT][4];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            float mx = max32_tree(sv[0][qt], sv[1][qt]);
            {
                auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
                mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
            }
            const float mnew = fmaxf(mrun[qt], mx);
            const float mc = (mnew == -INFINITY) ? 0.f : mnew * c;
            const float alpha = __builtin_amdgcn_exp2f(mrun[qt] * c - mc);
            mrun[qt] = mnew;
            if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) o[qt][dt] = o[qt][dt] * alpha;
                if (!ONES) lsum[qt] *= alpha;
            }
            float part = 0.f;
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                V8 f;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float pe = __builtin_amdgcn_exp2f(fmaf(sv[st >> 1][qt][8 * (st & 1) + e], c, -mc));
                    if (!ONES) part += pe;
                    f[e] = (T)pe;
                }
                pf[qt][st] = f;
            }
            if (!ONES) lsum[qt] += part;
        }

        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                const char* vp = sV + (16 * st + 4 * (g16 >> 1) + (l15 >> 2)) * VROW + (dt * 32 + 16 * (g16 & 1) + (l15 & 3) * 4) * 2;
                U128 vf;
                vf.d[0] = lds_read_tr16(vp);
                vf.d[1] = lds_read_tr16(vp + 8 * VROW);
                const V8 v8 = as_v8<T>(vf.u);
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) o[qt][dt] = mfma32(v8, pf[qt][st], o[qt][dt]);
            }
        __builtin_amdgcn_s_setprio(0);

#ifdef LDX_ATTN_NO_T14
        if (more) lstore(cur ^ 1);
#endif
        __syncthreads();
    }

    // finalize: denominator = O^T row D: d tile D / 32, register 4*((D%32)>>3) + (D%32 & 3) on the half ((D%32)>>2)&1 — D % 8 == 0,
    // so the register is 4*(dd>>3) and it is selected with static indices
    const int dd = D & 31, tD = D >> 5;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        float mine;
        if (ONES) {
            float cand = 0.f;
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                const float c0 = o[qt][dt][0], c1 = o[qt][dt][4], c2 = o[qt][dt][8], c3 = o[qt][dt][12];
                const float pick = (dd >> 3) == 0 ? c0 : ((dd >> 3) == 1 ? c1 : ((dd >> 3) == 2 ? c2 : c3));
                cand = (dt == tD) ? pick : cand;
            }
            mine = (h2 == ((dd >> 2) & 1)) ? cand : 0.f;
        } else {
            mine = lsum[qt];                          // this half's keys; the other half of the wave holds the rest
        }
        auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mine), __float_as_uint(mine), false, false);
        const float l = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
        const float inv = (l > 0.f) ? 1.0f / l : 0.f;
        const int q = q0 + qt * 32 + l31;
        if constexpr (NDT == 4 && !ONES) {
            // MX fp8 output (D = 128: the 4 tiles dt are the head's 4 blocks of 32; a block is split over the two half-waves)
            if (p.O8) {
                uint32_t sc = 0;
                char* o8 = (char*)p.O8 + ((long)b * p.Nq + q) * p.ldo8 + h * 128;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    float v[16], amax = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) { v[r] = to_f32(from_f32<T>(o[qt][dt][r] * inv)); amax = fmaxf(amax, fabsf(v[r])); }
                    auto am = __builtin_amdgcn_permlane32_swap(__float_as_uint(amax), __float_as_uint(amax), false, false);
                    amax = fmaxf(__uint_as_float(am[0]), __uint_as_float(am[1]));
                    const int e = mx_scale_e8m0(amax);
                    const float is = mx_inv_scale(e);
                    sc |= (uint32_t)e << (8 * dt);
                    if (q < p.Nq) {
#pragma unroll
                        for (int rq = 0; rq < 4; ++rq)
                            *(uint32_t*)(o8 + dt * 32 + 8 * rq + 4 * h2) = mx_pack4(v[4 * rq] * is, v[4 * rq + 1] * is, v[4 * rq + 2] * is, v[4 * rq + 3] * is);
                    }
                }
                if (q < p.Nq && h2 == 0) p.SO[(long)h * p.so_ld + (long)b * p.Nq + q] = sc;
                continue;
            }
        }
        if (q >= p.Nq) continue;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int d = dt * 32 + 8 * rq + 4 * h2;
                if (d < D)
                    *(uint2*)(Op + (long)q * p.ldo + d) = pack4<T>(o[qt][dt][4 * rq] * inv, o[qt][dt][4 * rq + 1] * inv, o[qt][dt][4 * rq + 2] * inv, o[qt][dt][4 * rq + 3] * inv);
            }
    }
}

template <typename T, int NKS, int NDT, int QT, bool ONES>
static void launch_attn32g(const AttnArgs& a, hipStream_t s) {
    constexpr int KROW = G32_KROW(NKS), VROW = G32_VROW(NDT);
    const size_t lds = 2 * AT_KV * (KROW + VROW);
    constexpr int QB = 128 * QT;
    dim3 grid(((a.Nq + QB - 1) / QB) * a.H * a.B);
    static const bool kpf = !(getenv("LDX_ATTN_KPF") && atoi(getenv("LDX_ATTN_KPF")) == 0);      // K fragment prefetch ring (round 3); 0: hipcc's just-in-time reads
    if (kpf) {
        static DevOnce once;
        set_dyn_lds(once, (const void*)attn32g_kernel<T, NKS, NDT, QT, ONES, true>, (int)lds);
        hipLaunchKernelGGL((attn32g_kernel<T, NKS, NDT, QT, ONES, true>), grid, dim3(256), lds, s, a);
    } else {
        static DevOnce once;
        set_dyn_lds(once, (const void*)attn32g_kernel<T, NKS, NDT, QT, ONES, false>, (int)lds);
        hipLaunchKernelGGL((attn32g_kernel<T, NKS, NDT, QT, ONES, false>), grid, dim3(256), lds, s, a);
    }
}
// head dims served by the generic 32x32x16 kernel (LDX_ATTN32G bit mask 1: D=80, 2: D=160, 4: D=128, 8: D=64).  Default 7: same-box
// step A/B 54.76 -> 55.34 it/s with D = 80 and 160; D = 128 (Flux, VALU denominator instead of a fifth d tile) 19.6 -> 18.4 ms of
// attention per forward; D = 64 only appears with a causal mask or a bias on this path.
static int attn32g_variant(const AttnArgs& a) {      // 0: not taken, else the head dim handled by a 32x32 instantiation
    static const int mask = getenv("LDX_ATTN32G") ? atoi(getenv("LDX_ATTN32G")) : 7;
    if (!mask || a.causal || a.bias) return 0;
    const long wg = (long)((a.Nq + 127) / 128) * a.H * a.B;
    static const long min_wg = getenv("LDX_ATTN32G_MINWG") ? atol(getenv("LDX_ATTN32G_MINWG")) : 16;      // 64 -> 16: +0.9 % on the 512^2 step (the 16x16 D = 160 instantiation spills)
    if (wg < min_wg) return 0;
    if ((mask & 1) && a.D == 80) return 80;
    if ((mask & 2) && a.D == 160) return 160;
    if ((mask & 4) && a.D == 128) return 128;
    if ((mask & 8) && a.D == 64) return 64;
    return 0;
}
bool attention_mx_out_ok(const AttnArgs& a) { return attn32g_variant(a) == 128; }
template <typename T>
static bool try_attn32g(const AttnArgs& a, hipStream_t s) {
    switch (attn32g_variant(a)) {
        case 80: launch_attn32g<T, 5, 3, 1, true>(a, s); return true;
        case 160: launch_attn32g<T, 10, 6, 1, true>(a, s); return true;
        case 128: launch_attn32g<T, 8, 4, 1, false>(a, s); return true;
        case 64: launch_attn32g<T, 4, 2, 1, false>(a, s); return true;
        default: return false;
    }
}

template <typename T, int KS, int DT, int QT>
static void launch_attn_q(const AttnArgs& a, hipStream_t s) {
    constexpr int STAGE = AT_KV * (AttnCfg<KS, DT>::KROWB + AttnCfg<KS, DT>::VROWB);
    constexpr int QB = 64 * QT;
    const size_t lds = 2 * STAGE;
    static DevOnce once;
    set_dyn_lds(once, (const void*)attn_kernel<T, KS, DT, QT>, (int)lds);
    dim3 grid(((a.Nq + QB - 1) / QB) * a.H * a.B);
    hipLaunchKernelGGL((attn_kernel<T, KS, DT, QT>), grid, dim3(256), lds, s, a);
}
template <typename T, int KS, int DT>
static void launch_attn_t(const AttnArgs& a, hipStream_t s) {
    // 64 queries per wave when the accumulators fit (DT <= 3) and the grid still fills the chip
    if constexpr (KS == 2 && DT == 3) {
        // D = 40 (SD1.5 level 0) on 32x32x16 MFMAs: +4.6 % in isolation, +2.2 % on the whole step; LDX_ATTN32=0 falls back
        static const int v32 = getenv("LDX_ATTN32") ? atoi(getenv("LDX_ATTN32")) : 1;
        if (v32 && !a.causal && !a.bias && a.D > 32 && a.D % 8 == 0 && (long)((a.Nq + 255) / 256) * a.H * a.B >= 512) { launch_attn32<T>(a, s); return; }
    }
    if constexpr (DT <= 3) {
        if ((long)((a.Nq + 255) / 256) * a.H * a.B >= 512 && !a.causal) { launch_attn_q<T, KS, DT, 4>(a, s); return; }
    }
    // D >= 144 (DT >= 10): two 16-query tiles per wave need 12 more VGPRs than the file has (40 B of scratch per lane, hipcc
    // -Rpass-analysis): one tile per wave there.  Only small grids / masked calls get here (attn32g takes the rest).
    if constexpr (DT >= 10) launch_attn_q<T, KS, DT, 1>(a, s);
    else launch_attn_q<T, KS, DT, 2>(a, s);
}

template <typename T>
static void launch_attn_d(const AttnArgs& a, hipStream_t s) {
    const int D = a.D;
    if (try_attn32g<T>(a, s)) return;
    // KS = ceil(D/32) contraction steps, DT = floor(D/16) + 1 output tiles (room for the ones column at d = D)
    if (D < 16) launch_attn_t<T, 1, 1>(a, s);
    else if (D < 32) launch_attn_t<T, 1, 2>(a, s);
    else if (D == 32) launch_attn_t<T, 1, 3>(a, s);
    else if (D < 48) launch_attn_t<T, 2, 3>(a, s);
    else if (D < 64) launch_attn_t<T, 2, 4>(a, s);
    else if (D == 64) launch_attn_t<T, 2, 5>(a, s);
    else if (D < 80) launch_attn_t<T, 3, 5>(a, s);
    else if (D < 96) launch_attn_t<T, 3, 6>(a, s);
    else if (D == 96) launch_attn_t<T, 3, 7>(a, s);
    else if (D < 128) launch_attn_t<T, 4, 8>(a, s);
    else if (D == 128) launch_attn_t<T, 4, 9>(a, s);
    else if (D < 160) launch_attn_t<T, 5, 10>(a, s);
    else launch_attn_t<T, 5, 11>(a, s);
}

// Dispatch switches of launch_attention, read ONCE (static initialisation) — the launch path itself never calls getenv.
// Experiments and tests that flip a switch inside one process call reload_dispatch_env() (C ABI: ldx_reload_env) after changing the environment.
struct AttnSwitches { bool pipe40, pipe128; long minwg40, minwg128; float thr; };
static AttnSwitches read_attn_switches() {
    AttnSwitches w;
    const char* e = getenv("LDX_ATTN_PIPE");
    const char* e128 = getenv("LDX_ATTN_PIPE128");
    const char* m = getenv("LDX_ATTN_PIPE_MINWG");
    const char* t = getenv("LDX_ATTN_PIPE_THR");
    w.pipe40 = !e || atoi(e) != 0;
    w.pipe128 = !e128 || atoi(e128) != 0;
    w.minwg40 = m ? atol(m) : 256;
    w.minwg128 = m ? atol(m) : 192;
    w.thr = t ? (float)atof(t) : __builtin_nanf("");          // NaN = the type's own rescale threshold
    return w;
}
static AttnSwitches g_attn_sw = read_attn_switches();
void reload_dispatch_env() { g_attn_sw = read_attn_switches(); }

// which kernel family launch_attention() takes for `a` (planner / profile labels): 3 attn512, 1 attn40p, 2 attn128p, 0 the generic dispatch (attn32g / attn32 / attn)
int attention_dispatch_class(const AttnArgs& a) {
    const AttnSwitches& w = g_attn_sw;
    if (attn512_ok(a)) return 3;
    if (w.pipe40 && attn_pipe_ok(a) && (long)(a.Nq / 256) * a.H * a.B >= w.minwg40) return 1;
    if (w.pipe128 && attn_pipe128_ok(a) && (long)(a.Nq / 256) * a.H * a.B >= w.minwg128) return 2;
    return 0;
}

void launch_attention(const AttnArgs& a, DType dt, hipStream_t s) {
    if (a.Nq <= 0 || a.B <= 0) return;
    const AttnSwitches& w = g_attn_sw;
    if (attn512_ok(a)) { launch_attn512(a, dt, s); return; }      // one head of D = 512 (VAE mid-block attention): attn512.hip
    // D = 40 self-attention of the large levels: the software-pipelined one-wave-per-SIMD kernel (attn_pipe.hip) when its 256-query workgroups
    // fill the chip.  LDX_ATTN_PIPE=0 restores attn32ap / attn32; LDX_ATTN_PIPE_THR=<x> overrides the rescale threshold (tests).
    if (w.pipe40 && attn_pipe_ok(a) && (long)(a.Nq / 256) * a.H * a.B >= w.minwg40) {
        launch_attn_pipe(a, dt, s, w.thr);
        return;
    }
    // D = 128 (Flux): the same pipeline (attn_pipe128.hip) when one round of its 256-query workgroups covers at least three quarters of the CUs.
    // O8 (MX fp8 output) needs attention_mx_out_ok(), i.e. the attn32g D = 128 variant enabled: the callers' test, unchanged.
    if (w.pipe128 && attn_pipe128_ok(a) && (long)(a.Nq / 256) * a.H * a.B >= w.minwg128) {
        launch_attn_pipe128(a, dt, s, w.thr);
        return;
    }
    if (dt == DT_BF16) launch_attn_d<__bf16>(a, s); else launch_attn_d<_Float16>(a, s);
}

}  // namespace ldx
