// Software-pipelined flash attention for D = 40 (SD1.5 level 0 self-attention: the step's largest kernel) — round 4.
// Reference call site: Attention.py:100-124 through AttentionMethods.py:107-150 (F.scaled_dot_product_attention, no mask).
//
// Same transposed 32x32x16 formulation, operand layouts and LDS strides as attn32_kernel (attention.hip):
//   S^T[key][q] = K . Q^T (contraction padded 40 -> 48), O^T[d][q] += V^T . P^T, lane l owns query l & 31 of a 32-query tile,
//   softmax denominator from the ones column of V at d = 40.
// What is different is the schedule, built on three measurements (profiles/ubench/README.md round 3, MI355X guide "Two waves per SIMD"):
// the matrix phase of one wave and the softmax phase of its SIMD partner do not overlap, a wave's OWN VALU work does issue in the
// shadow of its own MFMAs (about five single-issue fillers per 32-cycle 32x32x16 gap), and the kernel was bound by instructions per key.
//   * ONE wave per SIMD (256 threads, all 512 registers): 64 queries per wave, 256 per workgroup.
//   * Software pipeline over 64-key blocks: in slot t the matrix pipe runs QK^T(t+1) (12 MFMAs) and PV(t-1) (16 MFMAs) while the
//     VALU turns S'(t) into P(t) and takes the maximum of S'(t+1); S' and P are double-buffered in registers (loop unrolled by two).
//   * Lazy INTEGER reference maximum carried inside the QK^T contraction: the contraction is padded 40 -> 48 anyway, so K's LDS rows hold
//     the constant 1 at d = 40 and 41 and the Q fragment holds -m_ref of the lane's query there as a 16-bit hi / lo pair (exact for
//     |m_ref| < 6e4): the MFMA result already is S' = s - m_ref, and the softmax is exp2 / pack / max only — no per-element subtract or
//     scale (scale * log2(e) is folded into the Q fragments once; the engine passes scale = 1 / log2(e) with the softmax scale folded into
//     the q weights, for which this is the identity).  m_ref starts as ceil(max of block 0).  The maximum of S'(t+1) is taken late in
//     slot t, BEFORE any of its exponentials: only if it exceeds THR the reference is raised by an integer dl (rare path at the end of the
//     slot, all in place: O *= 2^-dl, P(t) by exponent subtraction — exact, P(t) <= 2^THR is finite by the previous check —
//     S'(t+1) -= dl, Q slots rewritten).  So P <= 2^THR always, which 16-bit P and fp32 accumulation hold without loss (the
//     denominator comes from the same rounded P through the ones column of V).
//   * Register classes are explicit: a 512-register wave has 256 arch VGPRs (all the VALU can address) and 256 AGPRs.  hipcc puts every
//     MFMA result into AGPRs at this budget and copies S' back and forth (1358 v_accvgpr moves per two slots in the first build), so the
//     QK^T MFMAs are inline asm with constraints: S' (MFMA -> VALU) in arch VGPRs, Q / K fragments in AGPRs (LDS reads and staging loads land
//     there directly).  The PV MFMAs stay builtins: O is only touched by the matrix pipe, so hipcc's AGPR home is the right one, and their
//     V^T / P operands may sit in either class.  The asm is opaque to the hazard recogniser: S' is only read by the VALU five or more
//     MFMA gaps after its last MFMA, or behind explicit s_nop runs (prologue).
//   * Issue order is pinned gap by gap (one MFMA, its fillers, sched_barrier(0)); the softmax pieces are asm blocks of fixed internal
//     order (a VALU read of a transcendental result needs a wait state: the pack of piece k - 1 sits in front of the exponentials of piece k).
//   * K / V tiles register-staged a slot ahead (buffer loads at the top of the slot, LDS stores in its second half), 2-deep rings, one
//     s_barrier per slot between the two matrix phases; fragment reads for the next slot are issued behind the MFMAs that last use the registers.
// Shapes taken: D = 40, Nq % 256 == 0, Mk % 128 == 0, Mk >= 256, no mask / bias; everything else stays on attn32* (attention.hip).
#include <stdlib.h>
#include <math.h>
#include <type_traits>
#include <utility>
#include "ldx_device.h"
#include "ldx_kernels.h"
#include "attn_pipe_common.h"

namespace ldx {

// ABL: timing ablations (wrong results; LDX_ATTN_PIPE_ABL, profiles/ubench): 1 no s_barrier, 2 no maximum, 8 no staging, 16 no fragment re-reads, 32 no exponentials
// max over the 64 keys of a block of ||k||_2 (x 1.002: rounding margin), one wave per (batch, head, block), lane = key.  With ||q c||_2 per query the
// main kernel gets  s - m_ref <= ||q c|| * knorm[block] - m_ref  for every score of the block without looking at the scores.
template <typename T>
__global__ __launch_bounds__(256) void attn_knorm_kernel(const AttnArgs p) {
    const int lane = threadIdx.x & 63, nblk = (p.Mk + 63) >> 6;
    const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= (long)p.B * p.H * nblk) return;
    const int blk = (int)(w % nblk), hb = (int)(w / nblk), h = hb % p.H, b = hb / p.H;
    const int key = blk * 64 + lane;
    float ss = 0.f;
    if (key < p.Mk) {
        const T* kp = (const T*)p.K + ((long)b * p.Mk + key) * p.ldk + h * p.D;
        for (int ch = 0; ch < p.D / 8; ++ch) {
            float f[8]; unpack8<T>(*(const uint4*)(kp + ch * 8), f);
#pragma unroll
            for (int e = 0; e < 8; ++e) ss = fmaf(f[e], f[e], ss);
        }
    }
    ss = wave_max(ss);
    if (lane == 0) p.knorm_ws[w] = sqrtf(ss) * 1.002f;
}

// Softmax pieces per MFMA gap (32 per slot).  A piece costs about 20 issue cycles (two quarter-rate exponentials and a pack), an LDS fragment read 8,
// a staging store 13, a staging load with its address arithmetic about 24; the tables give every gap a similar total.
template <bool KB> struct ApSched {
    static constexpr int BAR = 8;                    // the slot's barrier sits in front of gap BAR
    static constexpr int n(int g) {
        constexpr int kb[28] = {1, 1, 1, 1, 2, 1, 2, 1,  1, 1, 1, 1, 1, 1, 1, 1,  1, 1, 1, 1, 1, 1,  2, 1, 1, 1, 2, 1};
        constexpr int ex[28] = {1, 1, 1, 1, 2, 2, 1, 1,  1, 1, 1, 1, 1, 1, 1, 1,  1, 1, 1, 1, 1, 1, 1, 1,  1, 1, 2, 2};      // exact maximum: its blocks sit in gaps 16-25
        return KB ? kb[g] : ex[g];
    }
    static constexpr int first(int g) { int s = 0; for (int i = 0; i < g; ++i) s += n(i); return s; }      // first piece of gap g; first(28) == 32
};
static_assert(ApSched<true>::first(28) == 32 && ApSched<false>::first(28) == 32, "32 softmax pieces per slot");

// KB: key-block bound (AttnArgs::knorm_ws): the per-score maximum is taken only for blocks whose bound exceeds the threshold
template <typename T, int ABL, bool KB>
__global__ __launch_bounds__(256, 1) void attn40p_kernel(const AttnArgs p, const float thr) {
    constexpr int KROWB = 144, VROWB = 192, KVB = 64, D = 40, DCH = 5;
    constexpr int KBYTES = KVB * KROWB, VBYTES = KVB * VROWB;
    extern __shared__ __attribute__((aligned(16))) char smem[];       // K ring [2][KBYTES] | V ring [2][VBYTES]
    using V8 = typename Vec<T>::v8;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h2 = lane >> 5, l15 = lane & 15, g16 = lane >> 4;
    const int nqb = p.Nq >> 8;
    const int lin = xcd_remap(blockIdx.x, nqb * p.H * p.B);
    const int qblk = lin % nqb, hb = lin / nqb;
    const int h = hb % p.H, b = hb / p.H;
    const int q0 = qblk * 256 + wave * 64;
    const T* __restrict__ Qp = (const T*)p.Q + (long)b * p.Nq * p.ldq + h * D;
    const T* Kp = (const T*)p.K + (long)b * p.Mk * p.ldk + h * D;
    const T* Vp = (const T*)p.V + (long)b * p.Mk * p.ldv + h * D;
    T* __restrict__ Op = (T*)p.O + (long)b * p.Nq * p.ldo + h * D;
    const float c = p.scale * 1.44269504088896340736f;
    const int nblk = p.Mk >> 6;

    for (int i = tid; i < (2 * (KBYTES + VBYTES)) / 16; i += 256) *(uint4*)(smem + i * 16) = make_uint4(0, 0, 0, 0);
    __syncthreads();
    if (tid < 2 * KVB) {
        *(T*)(smem + 2 * KBYTES + (tid >> 6) * VBYTES + (tid & 63) * VROWB + D * 2) = (T)1.0f;       // ones column of V at d = D: row D of O^T = softmax denominator
        *(T*)(smem + (tid >> 6) * KBYTES + (tid & 63) * KROWB + D * 2) = (T)1.0f;                    // ones columns of K at d = D, D + 1: they meet the hi / lo
        *(T*)(smem + (tid >> 6) * KBYTES + (tid & 63) * KROWB + D * 2 + 2) = (T)1.0f;                // halves of -m_ref in the Q fragment
    }

    // Q fragments (B operand of QK^T): lane holds q = l31, d = 16 ks + 8 h2 .. +7, pre-multiplied by scale * log2(e) unless that is 1;
    // elements 0, 1 of qf[qt][2] on the h2 = 1 half (d = 40, 41) carry -m_ref
    ap_i32x4 qf[2][3];
    float qn[2] = {0.f, 0.f};                        // KB: ||q c||_2 of the lane's query (x 1.002)
    const bool unit = fabsf(c - 1.0f) < 1e-6f;
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int q = q0 + qt * 32 + l31;
        float ss = 0.f;
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            const int ch = 2 * ks + h2;
            uint4 u = make_uint4(0, 0, 0, 0);
            if (ch < DCH) u = *(const uint4*)(Qp + (long)q * p.ldq + ch * 8);
            V8 v = as_v8<T>(u);
            if (!unit) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (T)((float)v[e] * c);
            }
            if constexpr (KB) {
#pragma unroll
                for (int e = 0; e < 8; ++e) ss = fmaf((float)v[e], (float)v[e], ss);
            }
            ap_i32x4 w = ap_bits(v);
            asm volatile("" : "+a"(w));
            qf[qt][ks] = w;
        }
        if constexpr (KB) {
            auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(ss), __float_as_uint(ss), false, false);
            qn[qt] = sqrtf(__uint_as_float(sw[0]) + __uint_as_float(sw[1])) * 1.002f;
        }
    }
    const __amdgpu_buffer_rsrc_t rN = __builtin_amdgcn_make_buffer_rsrc((void*)(KB ? p.knorm_ws + (long)hb * nblk : nullptr), 0, KB ? nblk * 4 : 0, 0x00020000);

    // ---- staging: 640 16-byte chunks per slot (K tile 320 + V tile 320) over 256 threads in three rounds; the tile a round moves is wave-uniform
    //   round 0: K chunk tid;  round 1: wave 0: K chunk 256 + lane, waves 1-3: V chunk tid - 64;  round 2: waves 0, 1: V chunk tid + 192
    const bool r1k = wave == 0;
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    unsigned gofs[3], lofs[2][3];                    // lofs[ring slot][round]: LDS byte addresses
    {
        // round 2 on waves 2, 3 (no chunks left) repeats V chunks tid - 128, which round 1 also wrote: same bytes, no branch in the slot
        const int ck[3] = {tid, r1k ? 256 + tid : tid - 64, tid < 128 ? tid + 192 : tid - 128};
        const bool isk[3] = {true, r1k, false};
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int row = ck[i] / DCH, ch = ck[i] - row * DCH;
            gofs[i] = (unsigned)((row * (isk[i] ? p.ldk : p.ldv) + ch * 8) * 2);
            lofs[0][i] = lds_base + (unsigned)(isk[i] ? row * KROWB + ch * 16 : 2 * KBYTES + row * VROWB + ch * 16);
            lofs[1][i] = lofs[0][i] + (isk[i] ? KBYTES : VBYTES);
        }
    }
    const unsigned kstep = (unsigned)(KVB * p.ldk * 2), vstep = (unsigned)(KVB * p.ldv * 2);      // one batch of K / V is far below 4 GiB
    uint4 rsA[3], rsB[3];                            // two staging sets (named, statically indexed): the loads of slot t are stored in slot t + 1
    // tile indices are clamped to the last block: the tail slots re-stage it into ring slots nobody reads any more.  Buffer loads: the per-lane
    // chunk offset in voffset, the tile's byte offset in soffset (SALU only)
    const __amdgpu_buffer_rsrc_t rK = __builtin_amdgcn_make_buffer_rsrc((void*)Kp, 0, (int)(((long)(p.Mk - 1) * p.ldk + D) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rV = __builtin_amdgcn_make_buffer_rsrc((void*)Vp, 0, (int)(((long)(p.Mk - 1) * p.ldv + D) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t r1 = r1k ? rK : rV;
    const unsigned step1 = r1k ? kstep : vstep;
    auto gload1 = [&](uint4 (&rs)[3], int i, int kblk, int vblk) __attribute__((always_inline)) {
        kblk = min(kblk, nblk - 1); vblk = min(vblk, nblk - 1);
        const auto v = i == 0 ? __builtin_amdgcn_raw_buffer_load_b128(rK, gofs[0], kblk * kstep, 0)
                     : (i == 1 ? __builtin_amdgcn_raw_buffer_load_b128(r1, gofs[1], (r1k ? kblk : vblk) * step1, 0)
                               : __builtin_amdgcn_raw_buffer_load_b128(rV, gofs[2], vblk * vstep, 0));
        rs[i] = make_uint4(v[0], v[1], v[2], v[3]);
    };
    auto gload = [&](uint4 (&rs)[3], int kblk, int vblk) __attribute__((always_inline)) { gload1(rs, 0, kblk, vblk); gload1(rs, 1, kblk, vblk); gload1(rs, 2, kblk, vblk); };
    // the same loads from running byte offsets (the slot loop keeps them in SGPRs: add + clamp per stream instead of min + multiply per load)
    auto gload1o = [&](uint4 (&rs)[3], int i, unsigned kofs, unsigned vofs) __attribute__((always_inline)) {
        const auto v = i == 0 ? __builtin_amdgcn_raw_buffer_load_b128(rK, gofs[0], kofs, 0)
                     : (i == 1 ? __builtin_amdgcn_raw_buffer_load_b128(r1, gofs[1], r1k ? kofs : vofs, 0)
                               : __builtin_amdgcn_raw_buffer_load_b128(rV, gofs[2], vofs, 0));
        rs[i] = make_uint4(v[0], v[1], v[2], v[3]);
    };
    // asm with an AGPR data operand: the staging loads then land in AGPRs (with a VGPR home hipcc spilled them to AGPRs right behind the load,
    // i.e. vmcnt(0) three times per slot); hipcc still sees the load -> use dependency and places the vmcnt wait in front of the store
    auto lstore1 = [&](const uint4 (&rs)[3], int i, int slot) __attribute__((always_inline)) {
        const ap_i32x4 w = ap_bits(rs[i]);
        asm volatile("ds_write_b128 %0, %1" :: "v"(lofs[slot][i]), "a"(w) : "memory");
    };
    auto lstore = [&](const uint4 (&rs)[3], int slot) __attribute__((always_inline)) { lstore1(rs, 0, slot); lstore1(rs, 1, slot); lstore1(rs, 2, slot); };
    V8 kf[3][2];
    auto kread1 = [&](int ks, int kt, int slot) __attribute__((always_inline)) {
        kf[ks][kt] = as_v8<T>(*(const uint4*)(smem + slot * KBYTES + (kt * 32 + l31) * KROWB + (2 * ks + h2) * 16));
    };
    auto kread = [&](int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) { kread1(ks, 0, slot); kread1(ks, 1, slot); }
    };
    V8 vfA[4][2], vfB[4][2];                         // V^T fragments of blocks t - 1 (in use) and t (being read)
    auto vread1 = [&](V8 (&vf)[4][2], int st, int dt, int slot) __attribute__((always_inline)) {
        const char* vp = smem + 2 * KBYTES + slot * VBYTES + (16 * st + 4 * (g16 >> 1) + (l15 >> 2)) * VROWB + (dt * 32 + 16 * (g16 & 1) + (l15 & 3) * 4) * 2;
        vf[st][dt] = __builtin_bit_cast(V8, __builtin_shufflevector(ap_lds_read_tr16(vp), ap_lds_read_tr16(vp + 8 * VROWB), 0, 1, 2, 3));
    };

    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 o[2][2];                                  // [q tile][d tile of 32]; only the matrix pipe touches it in the loop: hipcc keeps it in AGPRs
    o[0][0] = zero16; o[0][1] = zero16; o[1][0] = zero16; o[1][1] = zero16;
    float mref[2] = {0.f, 0.f};                      // integer-valued
    f32x16 sA[2][2], sB[2][2];                       // S' = s - m_ref of blocks t (even t: sA) and t + 1
    ap_i32x4 pA[2][4], pB[2][4];                     // P^T of blocks t - 1 and t (B operand of PV), packed 16-bit pairs
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int st = 0; st < 4; ++st) { pA[qt][st] = (ap_i32x4){0, 0, 0, 0}; pB[qt][st] = pA[qt][st]; }
#pragma unroll
    for (int st = 0; st < 4; ++st) { vfA[st][0] = as_v8<T>(make_uint4(0, 0, 0, 0)); vfA[st][1] = vfA[st][0]; vfB[st][0] = vfA[st][0]; vfB[st][1] = vfA[st][0]; }

    // matrix work of a slot, one MFMA per call: QK^T i in [0, 12): (ks, kt, qt) = (i >> 2, (i >> 1) & 1, i & 1); PV j in [0, 16): (st, dt, qt)
    auto qk1 = [&](auto I, f32x16 (&s)[2][2]) __attribute__((always_inline)) {
        constexpr int i = decltype(I)::value, ks = i >> 2, kt = (i >> 1) & 1, qt = i & 1;
        if constexpr (ks == 0) ap_sacc0<T>(s[kt][qt], ap_bits(kf[0][kt]), qf[qt][0]);
        else if constexpr (ks == 1) ap_sacc<T>(s[kt][qt], ap_bits(kf[1][kt]), qf[qt][1]);
        else ap_sacc<T>(s[kt][qt], ap_bits(kf[2][kt]), qf[qt][2]);
    };
    auto pv1 = [&](auto J, ap_i32x4 (&pf)[2][4], V8 (&vf)[4][2]) __attribute__((always_inline)) {
        constexpr int j = decltype(J)::value, st = j >> 2, dt = (j >> 1) & 1, qt = j & 1;
        o[qt][dt] = mfma32(vf[st][dt], __builtin_bit_cast(V8, pf[qt][st]), o[qt][dt]);
    };
    // softmax piece k in [0, 32): dword k of P(t): (qt, st, w) = (k >> 4, (k >> 2) & 3, k & 3) = elements 2 w, 2 w + 1 of pf[qt][st], from
    // S' registers 8 (st & 1) + 2 w (+1) of tile (st >> 1, qt); its pack is issued with piece k + 1 (ap_piece), the last one by ap_pack
    constexpr int LAG = 3;                              // the pack trails by LAG pieces, the pieces rotate through LAG + 1 temporary pairs (see ap_piece)
    float ex[LAG + 1] = {0.f, 0.f, 0.f, 0.f}, ey[LAG + 1] = {0.f, 0.f, 0.f, 0.f};
    auto piece = [&](auto K, const f32x16 (&s)[2][2], ap_i32x4 (&pf)[2][4]) __attribute__((always_inline)) {
        constexpr int k = decltype(K)::value, qt = k >> 4, st = (k >> 2) & 3, w = k & 3;
        const float a = s[st >> 1][qt][8 * (st & 1) + 2 * w], bb = s[st >> 1][qt][8 * (st & 1) + 2 * w + 1];
        if constexpr (ABL & 32) { int r; asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(bb)); pf[qt][st][w] = r; }
        else if constexpr (k < LAG) ap_piece0(a, bb, ex[k], ey[k]);
        else { constexpr int j = k - LAG; pf[j >> 4][(j >> 2) & 3][j & 3] = ap_piece<T>(a, bb, ex[k % (LAG + 1)], ey[k % (LAG + 1)], ex[j % (LAG + 1)], ey[j % (LAG + 1)]); }
    };
    // maximum of S'(t+1): block m in [0, 8): (qt, kt, half) = (m >> 2, (m >> 1) & 1, m & 1), eight registers each
    float mx[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto maxblk = [&](auto M, const f32x16 (&s)[2][2]) __attribute__((always_inline)) {
        constexpr int m = decltype(M)::value, qt = m >> 2, kt = (m >> 1) & 1, r0 = 8 * (m & 1);
        const f32x16& v = s[kt][qt];
        mx[m] = ap_max8(v[r0], v[r0 + 1], v[r0 + 2], v[r0 + 3], v[r0 + 4], v[r0 + 5], v[r0 + 6], v[r0 + 7]);
    };
    auto bmax = [&](int qt) __attribute__((always_inline)) -> float {
        const float m = ap_max4(mx[4 * qt], mx[4 * qt + 1], mx[4 * qt + 2], mx[4 * qt + 3]);
        auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
        return ap_max2(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    };
    // set the (integer) reference of q tile qt: -m_ref = -(hi + lo), hi a multiple of `split`, both exact in T
    auto set_ref = [&](auto QT, float mnew) __attribute__((always_inline)) {
        constexpr int qt = decltype(QT)::value;
        mref[qt] = mnew;
        const float hi = truncf(mnew * (1.0f / ApT<T>::split)) * (float)ApT<T>::split, lo = mnew - hi;
        const T nh = (T)(-hi), nl = (T)(-lo);
        const unsigned bits = (unsigned)__builtin_bit_cast(unsigned short, nh) | ((unsigned)__builtin_bit_cast(unsigned short, nl) << 16);
        ap_i32x4 w = qf[qt][2];
        w[0] = h2 ? (int)bits : w[0];
        asm volatile("" : "+a"(w));                // the fragment's home stays an AGPR tuple (a VGPR home is copied over before every MFMA)
        qf[qt][2] = w;
    };

    // ---- prologue: K(0), K(1), V(0) staged; S(0) against m_ref = 0; m_ref := ceil(maximum of block 0); K(1) fragments in registers, K(2) in ring slot 0
    gload(rsA, 0, 0); __syncthreads(); lstore(rsA, 0);
    gload(rsA, 1, 0); lstore(rsA, 1);                // (V(0) goes to both V ring slots: a staging round always moves both tiles)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    kread(0);
    ap_for(ap_range<0, 12>(), [&](auto I) __attribute__((always_inline)) { qk1(I, sA); });
    kread(1);
    gload(rsA, 2, 0);
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");      // the asm MFMAs' results settle before the VALU reads them
    __syncthreads();                                 // every wave has read K(0) from ring slot 0
    lstore(rsA, 0);
    gload(rsB, 3, 1);                                // what slot 0 stores (K(3), V(1)): every slot stores the tiles loaded a slot earlier
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    ap_for(ap_range<0, 2>(), [&](auto QT) __attribute__((always_inline)) {
        constexpr int qt = decltype(QT)::value;
        float m = sA[0][qt][0];
#pragma unroll
        for (int r = 0; r < 16; ++r) m = fmaxf(m, fmaxf(sA[0][qt][r], sA[1][qt][r]));
        auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
        m = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        const float mr = fminf(fmaxf(ceilf(m), -60000.f), 60000.f);
#pragma unroll
        for (int r = 0; r < 16; ++r) { sA[0][qt][r] -= mr; sA[1][qt][r] -= mr; }
        set_ref(QT, mr);
    });

    // ---- slot t.  Matrix pipe: QK^T(t+1) (gaps 0-11), PV(t-1) (gaps 12-27).  VALU: S'(t) -> P(t) (32 pieces), maximum (or bound) of S'(t+1).
    // One barrier per slot: tiles stored in slot t - 1 (gaps 17-21) become readable behind it (V(t) fragments right after it, K(t+2) fragments
    // from gap 16), and this slot's stores overwrite ring slots whose last readers (slot t - 1, behind its barrier) every wave has passed before
    // it arrives.  Raw s_barrier + lgkmcnt(0) only: the staging loads issued at the top of the slot stay in flight across it.
    auto slot = [&](int t, f32x16 (&sc)[2][2], f32x16 (&sn)[2][2], ap_i32x4 (&pp)[2][4], ap_i32x4 (&pc)[2][4], V8 (&vfc)[4][2], V8 (&vfn)[4][2], uint4 (&rsl)[3], const uint4 (&rss)[3], auto PAR, unsigned kofs, unsigned vofs) __attribute__((always_inline)) {
        constexpr int par = decltype(PAR)::value;    // t & 1: ring slots are compile-time constants
        float kn = 0.f, bm0 = 0.f, bm1 = 0.f;
        // gap g: MFMA g (QK^T(t+1) for g < 12, PV(t-1) after), then its fillers from the schedule (ApSched): staging loads in gaps 0-2, the slot's
        // barrier behind gap BAR - 1, V(t) fragment reads (into the other fragment set) in the eight gaps from BAR on, K(t+2) fragment reads in gaps
        // 16-21 (this slot's QK^T MFMAs are issued), the staging stores in gaps 17 / 19 / 21, the 32 softmax pieces spread so that every gap costs
        // about the same (a gap cannot be shorter than its MFMA's 32 cycles, and the wave stalls in a gap that is lighter).
        auto gap = [&](auto G) __attribute__((always_inline)) {
            constexpr int g = decltype(G)::value;
            using S = ApSched<KB>;
            if constexpr (g == S::BAR) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if constexpr (!(ABL & 1)) __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (g < 12) qk1(G, sn); else pv1(ap_ic<g - 12>{}, pp, vfc);
            if constexpr (g < 3 && !(ABL & 8)) gload1o(rsl, g, kofs, vofs);                                              // stored in slot t + 1 (a whole slot of latency cover)
            if constexpr (KB && g == 3) kn = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rN, 0, min(t + 1, nblk - 1) * 4, 0));      // max ||k|| of block t + 1 (same for every lane)
            if constexpr (g >= S::BAR && g < S::BAR + 8 && !(ABL & 16)) vread1(vfn, (g - S::BAR) >> 1, (g - S::BAR) & 1, par);      // V(t), staged in slot t - 1
            if constexpr (g >= 16 && g < 22 && !(ABL & 16)) kread1((g - 16) >> 1, (g - 16) & 1, par);                  // K(t+2), staged in slot t - 1
            if constexpr ((g == 17 || g == 19 || g == 21) && !(ABL & 8)) lstore1(rss, (g - 17) >> 1, par ^ 1);          // K(t+3), V(t+1), loaded in slot t - 1
            ap_for(ap_range<S::first(g), S::first(g + 1)>(), [&](auto K) __attribute__((always_inline)) { piece(K, sc, pc); });
            if constexpr (!KB) {
                if constexpr (g >= 16 && g < 24 && !(ABL & 2)) maxblk(ap_ic<g - 16>{}, sn);                      // S'(t+1): its last MFMA issued in gap 11
                if constexpr (g == 24 && !(ABL & 2)) bm0 = bmax(0);
                if constexpr (g == 25 && !(ABL & 2)) bm1 = bmax(1);
            } else if constexpr (g == 24) {          // bound on S'(t+1) = s - m_ref from the norms: ||q c|| * max ||k|| - m_ref
                bm0 = fmaf(qn[0], kn, -mref[0]); bm1 = fmaf(qn[1], kn, -mref[1]);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        __builtin_amdgcn_sched_barrier(0);
        ap_for(ap_range<0, 28>(), gap);
        pc[1][3][1] = ap_pack<T>(ex[29 % (LAG + 1)], ey[29 % (LAG + 1)]);      // packs of pieces 29, 30, 31
        pc[1][3][2] = ap_pack<T>(ex[30 % (LAG + 1)], ey[30 % (LAG + 1)]);
        pc[1][3][3] = ap_pack<T>(ex[31 % (LAG + 1)], ey[31 % (LAG + 1)]);
        // rare path: some query's S'(t+1) exceeds thr: raise its reference by the integer dl BEFORE the exponentials of block t + 1 are taken
        auto raise_reference = [&]() __attribute__((always_inline)) {
            asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");      // PV(t-1) has landed in O
            ap_for(ap_range<0, 2>(), [&](auto QT) __attribute__((always_inline)) {
                constexpr int qt = decltype(QT)::value;
                const float mnew = fminf(mref[qt] + fmaxf(ceilf(qt ? bm1 : bm0), 0.f), 60000.f);
                const float dl = mnew - mref[qt];                                // integer >= 0
                { const float al = __builtin_amdgcn_exp2f(-dl); ap_scale_acc8<0>(o[qt][0], al); ap_scale_acc8<8>(o[qt][0], al); ap_scale_acc8<0>(o[qt][1], al); ap_scale_acc8<8>(o[qt][1], al); }
                const unsigned de = (unsigned)fminf(dl, (float)ApT<T>::maxdl) << ApT<T>::expsh;
                const unsigned de2 = de | (de << 16);
#pragma unroll
                for (int st = 0; st < 4; ++st)
#pragma unroll
                    for (int w = 0; w < 4; ++w) asm volatile("v_pk_sub_u16 %0, %0, %1 clamp" : "+v"(pc[qt][st][w]) : "v"(de2));      // P(t) *= 2^-dl (exponent field, saturating at 0)
#pragma unroll
                for (int r = 0; r < 16; ++r) { sn[0][qt][r] -= dl; sn[1][qt][r] -= dl; }
                set_ref(QT, mnew);
            });
        };
        if constexpr (KB) {
            // ONE branch on the common path: the bound does not prove the block safe for some query of this wave -> the exact maximum (what the other
            // variant does in every slot), and only inside that the threshold test proper.  (As two consecutive branches hipcc hoisted the rare path's
            // AGPR -> VGPR copies of O in front of the first one: 64 v_accvgpr_read per slot.)
            if (__builtin_amdgcn_ballot_w64(ap_max2(bm0, bm1) > thr) != 0) {
                asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
                ap_for(ap_range<0, 8>(), [&](auto M) __attribute__((always_inline)) { maxblk(M, sn); });
                bm0 = bmax(0); bm1 = bmax(1);
                if (__builtin_amdgcn_ballot_w64(ap_max2(bm0, bm1) > thr) != 0) raise_reference();
            }
        } else {
            if (__builtin_amdgcn_ballot_w64(ap_max2(bm0, bm1) > thr) != 0) raise_reference();
        }
    };
    // byte offsets of the tiles slot t loads: K(t + 4), V(t + 2), clamped to the last block
    const unsigned klast = (unsigned)(nblk - 1) * kstep, vlast = (unsigned)(nblk - 1) * vstep;
    unsigned kofs = min(4u * kstep, klast), vofs = min(2u * vstep, vlast);
    for (int t = 0; t < nblk; t += 2) {
        slot(t, sA, sB, pA, pB, vfA, vfB, rsA, rsB, ap_ic<0>{}, kofs, vofs);
        kofs = min(kofs + kstep, klast); vofs = min(vofs + vstep, vlast);
        slot(t + 1, sB, sA, pB, pA, vfB, vfA, rsB, rsA, ap_ic<1>{}, kofs, vofs);
        kofs = min(kofs + kstep, klast); vofs = min(vofs + vstep, vlast);
    }
    ap_for(ap_range<0, 16>(), [&](auto J) __attribute__((always_inline)) { pv1(J, pA, vfA); });      // P(nblk-1) (nblk even: written by the odd slot into pA) x V(nblk-1)
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");

    // ---- finalize: l from row D of O^T (ones column), normalise, store
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const float mine = (h2 == 0) ? o[qt][1][4] : 0.f;          // d = 40: tile 1, row 8 = register 4 of the h2 = 0 half
        auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mine), __float_as_uint(mine), false, false);
        const float l = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
        const float inv = (l > 0.f) ? 1.0f / l : 0.f;
        const int q = q0 + qt * 32 + l31;
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

// Measured and NOT kept (profiles/ubench/README.md round 4): the same pipeline at two waves per SIMD (eight waves x 32 queries, <= 256 registers, plain
// builtins): 863 us against 835 with QK^T / PV as phases, 963 us with the two MFMA kinds interleaved into four accumulator chains per wave — the two
// waves of a SIMD share its VALU issue port and matrix pipe, and what one gains the other loses.

// max ||k|| per 64-key block into a.knorm_ws (B * H * ceil(Mk / 64) floats): shared with attn_pipe128.hip
void launch_attn_knorm(const AttnArgs& a, DType dt, hipStream_t s) {
    const long waves = (long)a.B * a.H * ((a.Mk + 63) / 64);
    if (dt == DT_BF16) hipLaunchKernelGGL((attn_knorm_kernel<__bf16>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((attn_knorm_kernel<_Float16>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, a);
}

bool attn_pipe_ok(const AttnArgs& a) {
    return a.D == 40 && !a.causal && !a.bias && !a.O8 && a.Nq % 256 == 0 && a.Mk % 128 == 0 && a.Mk >= 256 &&
           a.ldq % 8 == 0 && a.ldk % 8 == 0 && a.ldv % 8 == 0 && a.ldo % 4 == 0;
}

template <typename T, int ABL = 0>
static void launch_attn40p(const AttnArgs& a, hipStream_t s, float thr) {
    const size_t lds = 2 * 64 * (144 + 192);
    dim3 grid((a.Nq / 256) * a.H * a.B);
    static const bool kb_off = getenv("LDX_ATTN_PIPE_KB") && atoi(getenv("LDX_ATTN_PIPE_KB")) == 0;      // experiment switch: exact maximum on every block
    if (a.knorm_ws && !kb_off && ABL == 0) {
        const long waves = (long)a.B * a.H * ((a.Mk + 63) / 64);
        hipLaunchKernelGGL((attn_knorm_kernel<T>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, a);
        hipLaunchKernelGGL((attn40p_kernel<T, ABL, true>), grid, dim3(256), lds, s, a, thr);
    } else {
        hipLaunchKernelGGL((attn40p_kernel<T, ABL, false>), grid, dim3(256), lds, s, a, thr);
    }
}

// thr_override: NaN = the type's default (tests force the rare path with small values)
void launch_attn_pipe(const AttnArgs& a, DType dt, hipStream_t s, float thr_override) {
    const bool ov = thr_override == thr_override;
#ifdef LDX_ATTN_ABLATE
    if (const char* e = getenv("LDX_ATTN_PIPE_ABL")) {
        const float thr = ApT<__bf16>::thr;
        switch (atoi(e)) {
            case 1: launch_attn40p<__bf16, 1>(a, s, thr); return;
            case 2: launch_attn40p<__bf16, 2>(a, s, thr); return;
            case 8: launch_attn40p<__bf16, 8>(a, s, thr); return;
            case 16: launch_attn40p<__bf16, 16>(a, s, thr); return;
            case 32: launch_attn40p<__bf16, 32>(a, s, thr); return;
            case 9: launch_attn40p<__bf16, 9>(a, s, thr); return;
            case 25: launch_attn40p<__bf16, 25>(a, s, thr); return;
            case 27: launch_attn40p<__bf16, 27>(a, s, thr); return;
            case 59: launch_attn40p<__bf16, 59>(a, s, thr); return;
            default: break;
        }
    }
#endif
    if (dt == DT_BF16) launch_attn40p<__bf16>(a, s, ov ? thr_override : ApT<__bf16>::thr);
    else launch_attn40p<_Float16>(a, s, ov ? fminf(thr_override, ApT<_Float16>::thr) : ApT<_Float16>::thr);
}

}  // namespace ldx
