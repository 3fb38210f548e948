// Software-pipelined flash attention for D = 128 (Flux DiT joint attention, 57 launches of B1 H24 N4352 per forward) — round 4.
// Reference call site: Flux.py:298-348 / 389-418 (attention() -> F.scaled_dot_product_attention on the RoPE'd q, k), no mask.
//
// The schedule of attn_pipe.hip (one wave per SIMD, 64 queries per wave; the matrix pipe runs QK^T of the NEXT keys and PV of the PREVIOUS ones while
// the VALU turns the current scores into P) carried to a head dim where the matrix work per 64 keys is 64 MFMAs instead of 28.  What differs from D = 40:
//   * the pipeline advances in HALF-slots of 32 keys (one 32-key tile x two q tiles): S and P buffers are 32 + 32 and 16 + 16 registers instead of twice
//     that (as 64-key slots the kernel needed 562 registers and spilled); per half-slot 16 QK^T MFMAs (8 k-steps x 2 q tiles) and 16 PV MFMAs
//     (2 key steps x 4 d tiles x 2 q tiles), one barrier per 64-key block.
//   * no spare contraction slots (128 = 8 x 16): S is the raw score and the softmax half-piece does exp2(fma(s, c, -m_ref)) itself, in fp32; the
//     denominator is summed on the VALU from the ROUNDED P (v_dot2_f32_{bf16,f16} with a pair of ones: one instruction per two keys, and the same rounded
//     values the PV MFMAs see) instead of a fifth d tile; the reference maximum is still lazy and integer-valued (raised by dl >= 0 at the end of a
//     half-slot, BEFORE any exponential of the next half, only when some score of it exceeds 2^THR: O, l and the packed P are scaled by 2^-dl, exactly).
//   * every K / V^T fragment is read just in time (three fragments ahead of its two MFMAs, rotating through four register sets) instead of holding a
//     block's fragments: what lives is S (64 VGPRs), P (32), O (128 AGPRs) and Q (64 AGPRs).
//   * the softmax is issued as half-pieces (one exponential each), ONE per MFMA gap.
//   * K / V rings are three deep: what block t stores (K(t+2), V(t+1)) is first read in block t + 1, behind that block's barrier, and its ring slots were
//     last read in block t - 1.  One staging register set: chunk i is stored and its register reloaded with the next tile's chunk in the same gap
//     (a full block of latency cover per load).
// Output: 16-bit O, or the MX fp8 bytes + one scale dword per (row, head) of attn32g's D = 128 epilogue (AttnArgs::O8, Flux fp8 mode).
// Shapes taken: D = 128, Nq % 256 == 0, Mk % 128 == 0, Mk >= 256, no mask / bias; everything else stays on attn32g (attention.hip).
#include <stdlib.h>
#include <math.h>
#include "ldx_device.h"
#include "ldx_kernels.h"
#include "attn_pipe_common.h"

namespace ldx {

// half-pieces of the softmax: A = first element of a pair + the pack of the pair LAG pieces back, B = second element + that pack's row sum.
// (Measured and not kept: scale folded into 16-bit Q fragments and -m_ref as the C operand of the first MFMA of a chain, i.e. no per-element fma:
// same 256 us per launch, and the rounded q c costs accuracy on large scores — rel-L2 1.1e-3 -> 1.4e-2 on the harness' spike case.)
template <typename T> __device__ __forceinline__ int ap128_half_a(float sa, float c, float nm, float& x, float px, float py) {
    int r;
    if constexpr (std::is_same<T, __bf16>::value) asm("v_fma_f32 %1, %2, %3, %4\n\tv_exp_f32 %1, %1\n\tv_cvt_pk_bf16_f32 %0, %5, %6" : "=&v"(r), "=&v"(x) : "v"(sa), "s"(c), "v"(nm), "v"(px), "v"(py));
    else asm("v_fma_f32 %1, %2, %3, %4\n\tv_exp_f32 %1, %1\n\tv_cvt_pk_f16_f32 %0, %5, %6" : "=&v"(r), "=&v"(x) : "v"(sa), "s"(c), "v"(nm), "v"(px), "v"(py));
    return r;
}
template <typename T> __device__ __forceinline__ void ap128_half_b(float sb, float c, float nm, float& y, float& l, int r, int ones) {
    if constexpr (std::is_same<T, __bf16>::value) asm("v_fma_f32 %0, %2, %3, %4\n\tv_exp_f32 %0, %0\n\tv_dot2_f32_bf16 %1, %5, %6, %1" : "=&v"(y), "+v"(l) : "v"(sb), "s"(c), "v"(nm), "v"(r), "s"(ones));
    else asm("v_fma_f32 %0, %2, %3, %4\n\tv_exp_f32 %0, %0\n\tv_dot2_f32_f16 %1, %5, %6, %1" : "=&v"(y), "+v"(l) : "v"(sb), "s"(c), "v"(nm), "v"(r), "s"(ones));
}
__device__ __forceinline__ void ap128_exp1(float s, float c, float nm, float& x) {
    asm("v_fma_f32 %0, %1, %2, %3\n\tv_exp_f32 %0, %0" : "=&v"(x) : "v"(s), "s"(c), "v"(nm));
}
// QK^T MFMAs of this kernel: K fragment (A operand) in arch VGPRs, Q fragment in AGPRs, S in arch VGPRs.  (With the K fragments in AGPRs as in attn_pipe.hip
// hipcc overlapped them with an O tile and saved / restored that tile around every block: 16 + 16 v_accvgpr copies.)
template <typename T> __device__ __forceinline__ void ap128_sacc0(f32x16& d, ap_i32x4 a, ap_i32x4 b) {
    if constexpr (std::is_same<T, __bf16>::value) asm volatile(AP_BF16 " %0, %1, %2, 0" : "=&v"(d) : "v"(a), "a"(b));
    else asm volatile(AP_F16 " %0, %1, %2, 0" : "=&v"(d) : "v"(a), "a"(b));
}
template <typename T> __device__ __forceinline__ void ap128_sacc(f32x16& d, ap_i32x4 a, ap_i32x4 b) {
    if constexpr (std::is_same<T, __bf16>::value) asm volatile(AP_BF16 " %0, %1, %2, %0" : "+v"(d) : "v"(a), "a"(b));
    else asm volatile(AP_F16 " %0, %1, %2, %0" : "+v"(d) : "v"(a), "a"(b));
}
// The dot product is opaque to hipcc's hazard recogniser inside an asm: it pads ONE wait state between this and a plain VALU read of l, and that is not
// enough on gfx950 — in the KB variant the copy of l for the loop exit sat right behind the last of these and read the OLD value: the last dword of P of
// the last half-slot was missing from the denominator of every query of q tile 1 (found as a probability mass of 0.12 on key 507 of 512).  Callers
// put an s_nop run behind the last row sum.
template <typename T> __device__ __forceinline__ void ap128_rowsum(float& l, int r, int ones) {
    if constexpr (std::is_same<T, __bf16>::value) asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(l) : "v"(r), "s"(ones));
    else asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(l) : "v"(r), "s"(ones));
}

// ABL: timing ablations (wrong results; LDX_ATTN_PIPE_ABL with -DLDX_ATTN_ABLATE): 1 no s_barrier, 2 no maximum, 4 no MFMAs, 8 no staging, 16 no fragment reads, 32 no softmax
template <typename T, int ABL, bool KB>
__global__ __launch_bounds__(256, 1) void attn128p_kernel(const AttnArgs p, const float thr) {
    constexpr int D = 128, KVB = 64, KROWB = 272, VROWB = 320;      // row strides: conflict-free b128 (K) and transposing b64 (V) fragment reads
    constexpr int KBYTES = KVB * KROWB, VBYTES = KVB * VROWB, VBASE = 3 * KBYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];       // K ring [3][KBYTES] | V ring [3][VBYTES]
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
    const int ones = std::is_same<T, __bf16>::value ? 0x3f803f80 : 0x3c003c00;

    for (int i = tid; i < (3 * (KBYTES + VBYTES)) / 16; i += 256) *(uint4*)(smem + i * 16) = make_uint4(0, 0, 0, 0);      // V ring 2 is "V(-1)" of slot 0

    // Q fragments (B operand of QK^T): lane holds q = l31, d = 16 ks + 8 h2 .. +7 (raw: the scale is applied with the reference in the softmax fma)
    ap_i32x4 qf[2][8];
    float qnc[2] = {0.f, 0.f};                       // KB: ||q c||_2 of the lane's queries (x 1.002)
    {
        uint4 qu[2][8];                              // all sixteen loads in flight before the first is pinned (an asm use waits for its load)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) qu[qt][ks] = *(const uint4*)(Qp + (long)(q0 + qt * 32 + l31) * p.ldq + (2 * ks + h2) * 8);
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                ap_i32x4 w = ap_bits(qu[qt][ks]);
                asm volatile("" : "+a"(w));          // AGPR home (see attn_pipe.hip)
                qf[qt][ks] = w;
            }
        if constexpr (KB) {
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                float ss = 0.f;
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    const V8 v = as_v8<T>(qu[qt][ks]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) ss = fmaf((float)v[e], (float)v[e], ss);
                }
                auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(ss), __float_as_uint(ss), false, false);
                qnc[qt] = sqrtf(__uint_as_float(sw[0]) + __uint_as_float(sw[1])) * 1.002f * c;      // ||q|| c (rounded up): s c <= ||q|| c ||k||
            }
        }
    }
    const __amdgpu_buffer_rsrc_t rN = __builtin_amdgcn_make_buffer_rsrc((void*)(KB ? p.knorm_ws + (long)hb * nblk : nullptr), 0, KB ? nblk * 4 : 0, 0x00020000);

    // ---- staging: a 64 x 256 B tile is 1024 16-byte chunks = four per thread (rows srow + 16 i, chunk sch); chunks 0-3 K, 4-7 V
    const int srow = tid >> 4, sch = tid & 15;
    const unsigned gofK = (unsigned)((srow * p.ldk + sch * 8) * 2), gofV = (unsigned)((srow * p.ldv + sch * 8) * 2);
    const unsigned rstepK = (unsigned)(16 * p.ldk * 2), rstepV = (unsigned)(16 * p.ldv * 2);
    const unsigned kstep = (unsigned)(KVB * p.ldk * 2), vstep = (unsigned)(KVB * p.ldv * 2);
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned lK = lds_base + (unsigned)(srow * KROWB + sch * 16), lV = lds_base + (unsigned)(VBASE + srow * VROWB + sch * 16);
    const __amdgpu_buffer_rsrc_t rK = __builtin_amdgcn_make_buffer_rsrc((void*)Kp, 0, (int)(((long)(p.Mk - 1) * p.ldk + D) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rV = __builtin_amdgcn_make_buffer_rsrc((void*)Vp, 0, (int)(((long)(p.Mk - 1) * p.ldv + D) * 2), 0x00020000);
    uint4 rs[8];
    auto gload1 = [&](auto I, int kblk, int vblk) __attribute__((always_inline)) {      // tile indices clamp to the last block (the tail re-stages it into ring slots nobody reads)
        constexpr int i = decltype(I)::value;
        kblk = min(kblk, nblk - 1); vblk = min(vblk, nblk - 1);
        const auto v = i < 4 ? __builtin_amdgcn_raw_buffer_load_b128(rK, gofK, kblk * kstep + i * rstepK, 0)
                             : __builtin_amdgcn_raw_buffer_load_b128(rV, gofV, vblk * vstep + (i - 4) * rstepV, 0);
        rs[i] = make_uint4(v[0], v[1], v[2], v[3]);
    };
    // the same loads from running byte offsets of the two tiles (kept in SGPRs by the block loop: add + clamp per block instead of min + multiply per load)
    auto gload1o = [&](auto I, unsigned kofs, unsigned vofs) __attribute__((always_inline)) {
        constexpr int i = decltype(I)::value;
        const auto v = i < 4 ? __builtin_amdgcn_raw_buffer_load_b128(rK, gofK, kofs + i * rstepK, 0)
                             : __builtin_amdgcn_raw_buffer_load_b128(rV, gofV, vofs + (i - 4) * rstepV, 0);
        rs[i] = make_uint4(v[0], v[1], v[2], v[3]);
    };
    auto lstore1 = [&](auto I, int kring, int vring) __attribute__((always_inline)) {
        constexpr int i = decltype(I)::value;
        const ap_i32x4 w = ap_bits(rs[i]);
        const unsigned a = i < 4 ? lK + (unsigned)(kring * KBYTES) : lV + (unsigned)(vring * VBYTES);
        asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(a), "a"(w), "i"((i & 3) * 16 * (i < 4 ? KROWB : VROWB)) : "memory");
    };
    auto gload_all = [&](int kblk, int vblk) __attribute__((always_inline)) { ap_for(ap_range<0, 8>(), [&](auto I) __attribute__((always_inline)) { gload1(I, kblk, vblk); }); };
    auto lstore_all = [&](int kring, int vring) __attribute__((always_inline)) { ap_for(ap_range<0, 8>(), [&](auto I) __attribute__((always_inline)) { lstore1(I, kring, vring); }); };

    // ---- fragments, read just in time: K fragment ks of a 32-key half (32 keys x 16 d, A operand of QK^T), V^T fragment f = 4 st + dt (32 d x 16 keys)
    const int lane_k = l31 * KROWB + h2 * 16;
    const int lane_v = VBASE + (4 * (g16 >> 1) + (l15 >> 2)) * VROWB + (16 * (g16 & 1) + (l15 & 3) * 4) * 2;
    V8 kfr[4], vfr[4];
    auto vread = [&](auto F, const char* vp) __attribute__((always_inline)) {
        constexpr int f = decltype(F)::value, st = f >> 2, dt = f & 3;
        const char* a = vp + st * 16 * VROWB + dt * 64;
        vfr[f & 3] = __builtin_bit_cast(V8, __builtin_shufflevector(ap_lds_read_tr16(a), ap_lds_read_tr16(a + 8 * VROWB), 0, 1, 2, 3));
    };

    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 o[2][4];                                  // [q tile][d tile of 32]: only the matrix pipe touches it in the loop (AGPRs)
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[qt][dt] = zero16;
    float nm[2] = {0.f, 0.f}, lsum[2] = {0.f, 0.f};  // nm = -m_ref (scaled units, integer-valued); lsum: this half-wave's keys
    f32x16 sX[2], sY[2];                             // raw S of the 32-key halves u (even u: sX) and u + 1, [q tile]
    ap_i32x4 pQ[2][2], pR[2][2];                     // P^T of halves u (even u: pQ) and u - 1, [q tile][16-key step]: B operands of PV, packed 16-bit pairs
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int st = 0; st < 2; ++st) { pQ[qt][st] = (ap_i32x4){0, 0, 0, 0}; pR[qt][st] = pQ[qt][st]; }
#pragma unroll
    for (int f = 0; f < 4; ++f) { vfr[f] = as_v8<T>(make_uint4(0, 0, 0, 0)); kfr[f] = vfr[f]; }

    // matrix work of a half-slot, one MFMA per call: QK^T i in [0, 16): K fragment ks = i >> 1, q tile i & 1; PV j in [0, 16): V^T fragment j >> 1
    // (16-key step (j >> 3) of the half, d tile (j >> 1) & 3), q tile j & 1
    auto qk1 = [&](auto I, f32x16 (&s)[2]) __attribute__((always_inline)) {
        constexpr int i = decltype(I)::value, ks = i >> 1, qt = i & 1;
        if constexpr (ks == 0) ap128_sacc0<T>(s[qt], ap_bits(kfr[0]), qf[qt][0]);
        else ap128_sacc<T>(s[qt], ap_bits(kfr[ks & 3]), qf[qt][ks]);
    };
    auto pv1 = [&](auto J, ap_i32x4 (&pf)[2][2]) __attribute__((always_inline)) {
        constexpr int j = decltype(J)::value, f = j >> 1, st = f >> 2, dt = f & 3, qt = j & 1;
        o[qt][dt] = mfma32(vfr[f & 3], __builtin_bit_cast(V8, pf[qt][st]), o[qt][dt]);
    };
    // softmax piece k in [0, 16) of a half: dword k of its P: (qt, st, w) = (k >> 3, (k >> 2) & 1, k & 3) = elements 2 w, 2 w + 1 of pf[qt][st], from S
    // registers 8 st + 2 w (+1) of tile qt.  Half-piece A: first element, and the pack of piece k - LAG (a VALU read of a fresh transcendental
    // result needs wait states); half-piece B: second element, and that pack's contribution to the row sum.
    constexpr int LAG = 3;
    float ex[LAG + 1] = {0.f, 0.f, 0.f, 0.f}, ey[LAG + 1] = {0.f, 0.f, 0.f, 0.f};
    int rlast = 0;
    auto half = [&](auto HK, const f32x16 (&s)[2], ap_i32x4 (&pf)[2][2]) __attribute__((always_inline)) {
        constexpr int hk = decltype(HK)::value, k = hk >> 1, qt = k >> 3, st = (k >> 2) & 1, w = k & 3, j = k - LAG;
        const float v = s[qt][8 * st + 2 * w + (hk & 1)];
        if constexpr ((hk & 1) == 0) {
            if constexpr (k < LAG) ap128_exp1(v, c, nm[qt], ex[k]);
            else { rlast = ap128_half_a<T>(v, c, nm[qt], ex[k % (LAG + 1)], ex[j % (LAG + 1)], ey[j % (LAG + 1)]); pf[j >> 3][(j >> 2) & 1][j & 3] = rlast; }
        } else {
            if constexpr (k < LAG) ap128_exp1(v, c, nm[qt], ey[k]);
            else ap128_half_b<T>(v, c, nm[qt], ey[k % (LAG + 1)], lsum[j >> 3], rlast, ones);
        }
    };
    // maximum of the next half's S: block m in [0, 4): (qt, register half) = (m >> 1, m & 1), eight registers each
    float mx[4] = {0.f, 0.f, 0.f, 0.f};
    auto maxblk = [&](auto M, const f32x16 (&s)[2]) __attribute__((always_inline)) {
        constexpr int m = decltype(M)::value, qt = m >> 1, r0 = 8 * (m & 1);
        const f32x16& v = s[qt];
        mx[m] = ap_max8(v[r0], v[r0 + 1], v[r0 + 2], v[r0 + 3], v[r0 + 4], v[r0 + 5], v[r0 + 6], v[r0 + 7]);
    };
    auto bmax = [&](int qt) __attribute__((always_inline)) -> float {      // max over the half of s c - m_ref, for the lane's query
        const float m = ap_max2(mx[2 * qt], mx[2 * qt + 1]);
        auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
        return fmaf(ap_max2(__uint_as_float(sw[0]), __uint_as_float(sw[1])), c, nm[qt]);
    };

    // ---- prologue: K(0), K(1) -> K ring 0, 1; V(0) -> V ring 0; K(2), V(1) in the staging registers; S of half 0; m_ref := ceil(its maximum);
    // fragments 0..2 of half 1 (K(0), keys 32-63)
    __syncthreads();
    gload_all(0, 0); lstore_all(0, 0);
    gload_all(1, 0); lstore_all(1, 0);
    gload_all(2, 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    {
        const char* kp = smem + lane_k;
        ap_for(ap_range<0, 8>(), [&](auto F) __attribute__((always_inline)) {
            constexpr int f = decltype(F)::value;
            kfr[f & 3] = as_v8<T>(*(const uint4*)(kp + f * 32));
            qk1(ap_ic<2 * f>{}, sX); qk1(ap_ic<2 * f + 1>{}, sX);
        });
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");      // the asm MFMAs' results settle before the VALU reads them
        ap_for(ap_range<0, 2>(), [&](auto QT) __attribute__((always_inline)) {
            constexpr int qt = decltype(QT)::value;
            float m = sX[qt][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) m = fmaxf(m, sX[qt][r]);
            auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
            m = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
            nm[qt] = -fminf(fmaxf(ceilf(m * c), -1e30f), 1e30f);
        });
        ap_for(ap_range<0, 3>(), [&](auto F) __attribute__((always_inline)) { kfr[decltype(F)::value] = as_v8<T>(*(const uint4*)(kp + 32 * KROWB + decltype(F)::value * 32)); });
    }

    // ---- block t (ring indices r0 = t % 3, r1 = (t + 1) % 3, r2 = (t + 2) % 3), two half-slots u = 2 t + PAR of 32 MFMA gaps each:
    //   matrix pipe   PAR 0: QK^T of keys 32-63 of block t (K ring r0) -> sn;   PV of keys 32-63 of block t - 1 (V ring r2) with pp
    //                 PAR 1: QK^T of keys 0-31 of block t + 1 (K ring r1) -> sn; PV of keys 0-31 of block t (V ring r0) with pp
    //   VALU          sc -> pc, one half-piece per gap; maximum of sn in gaps 21-31; the reference is raised (rarely) at the end of the half-slot,
    //                 before any exponential of sn is taken
    //   staging       K(t+2) -> K ring r2, V(t+1) -> V ring r1 (their last readers ran in block t - 1, in front of this block's barrier); the register
    //                 of a stored chunk is reloaded with K(t+3) / V(t+2) right away
    auto halfslot = [&](auto PARC, int t, f32x16 (&sc)[2], f32x16 (&sn)[2], ap_i32x4 (&pp)[2][2], ap_i32x4 (&pc)[2][2], int r0, int r1, int r2, unsigned kofs, unsigned vofs) __attribute__((always_inline)) {
        constexpr int PAR = decltype(PARC)::value;
        const char* kp = smem + (PAR ? r1 : r0) * KBYTES + (1 - PAR) * 32 * KROWB + lane_k;      // this half-slot's keys
        const char* kpn = smem + r1 * KBYTES + PAR * 32 * KROWB + lane_k;                         // the next half-slot's: fragments 0..2 for its first MFMAs
        const char* vp = smem + (PAR ? r0 : r2) * VBYTES + (1 - PAR) * 32 * VROWB + lane_v;
        float bm0 = 0.f, bm1 = 0.f, kn = 0.f;
        auto gap = [&](auto G) __attribute__((always_inline)) {
            constexpr int g = decltype(G)::value;
            if constexpr (g == 0 && PAR == 0) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if constexpr (!(ABL & 1)) __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (!(ABL & 4)) { if constexpr (g < 16) qk1(G, sn); else pv1(ap_ic<g - 16>{}, pp); }
            if constexpr ((g & 1) == 0 && g <= 8 && !(ABL & 16)) kfr[(g / 2 + 3) & 3] = as_v8<T>(*(const uint4*)(kp + (g / 2 + 3) * 32));
            if constexpr ((g & 1) == 0 && g >= 10 && g <= 24 && !(ABL & 16)) vread(ap_ic<(g - 10) / 2>{}, vp);
            if constexpr ((g == 26 || g == 28 || g == 30) && !(ABL & 16)) kfr[(g - 26) / 2] = as_v8<T>(*(const uint4*)(kpn + ((g - 26) / 2) * 32));
            if constexpr ((g & 7) == 5 && !(ABL & 8)) { lstore1(ap_ic<4 * PAR + (g >> 3)>{}, r2, r1); gload1o(ap_ic<4 * PAR + (g >> 3)>{}, kofs, vofs); }
            if constexpr (KB && g == 3) kn = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rN, 0, min(t + PAR, nblk - 1) * 4, 0));      // max ||k|| of the block sn belongs to
            if constexpr (!(ABL & 32)) half(G, sc, pc);
            if constexpr (!KB) {
                if constexpr ((g == 21 || g == 23 || g == 25 || g == 27) && !(ABL & 2)) maxblk(ap_ic<(g - 21) / 2>{}, sn);      // its last MFMA issued in gap 15
                if constexpr (g == 29 && !(ABL & 2)) bm0 = bmax(0);
                if constexpr (g == 31 && !(ABL & 2)) bm1 = bmax(1);
            } else if constexpr (g == 29) {          // bound on s c - m_ref from the norms (Cauchy-Schwarz), no look at the scores
                bm0 = fmaf(qnc[0], kn, nm[0]); bm1 = fmaf(qnc[1], kn, nm[1]);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        __builtin_amdgcn_sched_barrier(0);
        ap_for(ap_range<0, 32>(), gap);
        ap_for(ap_range<16 - LAG, 16>(), [&](auto J) __attribute__((always_inline)) {      // packs and row sums of the last LAG pieces
            constexpr int j = decltype(J)::value;
            const int r = ap_pack<T>(ex[j % (LAG + 1)], ey[j % (LAG + 1)]);
            pc[j >> 3][(j >> 2) & 1][j & 3] = r;
            ap128_rowsum<T>(lsum[j >> 3], r, ones);
        });
        asm volatile("s_nop 7" ::: "memory");          // the dot2 results land before anything reads lsum (see ap128_rowsum)
        // rare path: some query's score in sn exceeds 2^thr over the reference: raise it by the integer dl
        auto raise_reference = [&]() __attribute__((always_inline)) {
            asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");      // the PV MFMAs have landed in O
            ap_for(ap_range<0, 2>(), [&](auto QT) __attribute__((always_inline)) {
                constexpr int qt = decltype(QT)::value;
                const float dl = fminf(fmaxf(ceilf(qt ? bm1 : bm0), 0.f), 1e30f);      // integer >= 0
                const float al = __builtin_amdgcn_exp2f(-dl);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) { ap_scale_acc8<0>(o[qt][dt], al); ap_scale_acc8<8>(o[qt][dt], al); }
                lsum[qt] *= al;
                const unsigned de = (unsigned)fminf(dl, (float)ApT<T>::maxdl) << ApT<T>::expsh;
                const unsigned de2 = de | (de << 16);
#pragma unroll
                for (int st = 0; st < 2; ++st)
#pragma unroll
                    for (int w = 0; w < 4; ++w) asm volatile("v_pk_sub_u16 %0, %0, %1 clamp" : "+v"(pc[qt][st][w]) : "v"(de2));      // this half's P *= 2^-dl (exponent field, saturating at 0)
                nm[qt] -= dl;
            });
        };
        if constexpr (KB) {
            // ONE branch on the common path: the bound does not prove this half safe for some query of the wave -> the exact maximum (what the other
            // variant does in every half-slot), and only inside that the threshold test proper (see attn_pipe.hip)
            if (__builtin_amdgcn_ballot_w64(ap_max2(bm0, bm1) > thr) != 0) {
                asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
                ap_for(ap_range<0, 4>(), [&](auto M) __attribute__((always_inline)) { maxblk(M, sn); });
                bm0 = bmax(0); bm1 = bmax(1);
                if (__builtin_amdgcn_ballot_w64(ap_max2(bm0, bm1) > thr) != 0) raise_reference();
            }
        } else {
            if (__builtin_amdgcn_ballot_w64(ap_max2(bm0, bm1) > thr) != 0) raise_reference();
        }
    };
    int r0 = 0, r1 = 1, r2 = 2;
    const unsigned klast = (unsigned)(nblk - 1) * kstep, vlast = (unsigned)(nblk - 1) * vstep;
    unsigned kofs = min(3u * kstep, klast), vofs = min(2u * vstep, vlast);      // block t loads K(t + 3), V(t + 2), clamped to the last block
    for (int t = 0; t < nblk; ++t) {
        halfslot(ap_ic<0>{}, t, sX, sY, pR, pQ, r0, r1, r2, kofs, vofs);
        halfslot(ap_ic<1>{}, t, sY, sX, pQ, pR, r0, r1, r2, kofs, vofs);
        const int x = r0; r0 = r1; r1 = r2; r2 = x;
        kofs = min(kofs + kstep, klast); vofs = min(vofs + vstep, vlast);
    }
    // keys 32-63 of block nblk - 1 (pR) x V(nblk-1), stored in block nblk - 2; r2 = (nblk - 1) % 3 here
    {
        const char* vp = smem + r2 * VBYTES + 32 * VROWB + lane_v;
        ap_for(ap_range<0, 8>(), [&](auto F) __attribute__((always_inline)) {
            constexpr int f = decltype(F)::value;
            vread(F, vp);
            pv1(ap_ic<2 * f>{}, pR); pv1(ap_ic<2 * f + 1>{}, pR);
        });
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");

    // ---- finalize: l = both half-waves' row sums, normalise, store (16-bit, or MX fp8 + E8M0 scales: the four d tiles are the head's four blocks of 32).
    // The O^T accumulator layout gives a lane 8 bytes (4 as fp8) of 32 different rows per store; the wave transposes its 64 x 128 tile through the
    // (now idle) rings instead and writes whole rows: 16 bytes per lane, 256 (128) contiguous bytes per row (same launch time as the direct
    // stores in the harness, half the store instructions).
    __syncthreads();                                 // every wave is done with the rings
    constexpr int EROW = 272, EROW8 = 144;
    char* ep = smem + wave * (64 * EROW);
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(lsum[qt]), __float_as_uint(lsum[qt]), false, false);
        const float l = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
        const float inv = (l > 0.f) ? 1.0f / l : 0.f;
        const int q = q0 + qt * 32 + l31;
        if (p.O8) {
            uint32_t sc = 0;
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
#pragma unroll
                for (int rq = 0; rq < 4; ++rq)
                    *(uint32_t*)(ep + (qt * 32 + l31) * EROW8 + dt * 32 + 8 * rq + 4 * h2) = mx_pack4(v[4 * rq] * is, v[4 * rq + 1] * is, v[4 * rq + 2] * is, v[4 * rq + 3] * is);
            }
            if (h2 == 0) p.SO[(long)h * p.so_ld + (long)b * p.Nq + q] = sc;
            continue;
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq)
                *(uint2*)(ep + (qt * 32 + l31) * EROW + (dt * 32 + 8 * rq + 4 * h2) * 2) = pack4<T>(o[qt][dt][4 * rq] * inv, o[qt][dt][4 * rq + 1] * inv, o[qt][dt][4 * rq + 2] * inv, o[qt][dt][4 * rq + 3] * inv);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // wave-local exchange: the wave's own writes have landed
    if (p.O8) {
        char* o8 = (char*)p.O8 + ((long)b * p.Nq + q0) * p.ldo8 + h * 128;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = i * 8 + (lane >> 3), ch = lane & 7;
            *(uint4*)(o8 + (long)row * p.ldo8 + ch * 16) = *(const uint4*)(ep + row * EROW8 + ch * 16);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = i * 4 + (lane >> 4), ch = lane & 15;
            *(uint4*)(Op + (long)(q0 + row) * p.ldo + ch * 8) = *(const uint4*)(ep + row * EROW + ch * 16);
        }
    }
}

bool attn_pipe128_ok(const AttnArgs& a) {
    return a.D == 128 && !a.causal && !a.bias && a.Nq % 256 == 0 && a.Mk % 128 == 0 && a.Mk >= 256 &&
           a.ldq % 8 == 0 && a.ldk % 8 == 0 && a.ldv % 8 == 0 && (a.O8 ? (a.ldo8 % 16 == 0 && ((uintptr_t)a.O8 & 15) == 0) : (a.ldo % 8 == 0 && ((uintptr_t)a.O & 15) == 0));
}

template <typename T, int ABL = 0>
static void launch_attn128p(const AttnArgs& a, hipStream_t s, float thr) {
    const size_t lds = 3 * 64 * (272 + 320);
    const dim3 grid((a.Nq / 256) * a.H * a.B);
    static const bool kb_off = getenv("LDX_ATTN_PIPE_KB") && atoi(getenv("LDX_ATTN_PIPE_KB")) == 0;      // experiment switch: exact maximum on every half-slot
    if (a.knorm_ws && !kb_off && ABL == 0) {
        launch_attn_knorm(a, std::is_same<T, __bf16>::value ? DT_BF16 : DT_F16, s);
        static DevOnce once;
        set_dyn_lds(once, (const void*)attn128p_kernel<T, ABL, true>, (int)lds);
        hipLaunchKernelGGL((attn128p_kernel<T, ABL, true>), grid, dim3(256), lds, s, a, thr);
    } else {
        static DevOnce once;
        set_dyn_lds(once, (const void*)attn128p_kernel<T, ABL, false>, (int)lds);
        hipLaunchKernelGGL((attn128p_kernel<T, ABL, false>), grid, dim3(256), lds, s, a, thr);
    }
}

void launch_attn_pipe128(const AttnArgs& a, DType dt, hipStream_t s, float thr_override) {
    const bool ov = thr_override == thr_override;
#ifdef LDX_ATTN_ABLATE
    if (const char* e = getenv("LDX_ATTN_PIPE_ABL")) {
        const float thr = ApT<__bf16>::thr;
        switch (atoi(e)) {
            case 1: launch_attn128p<__bf16, 1>(a, s, thr); return;
            case 2: launch_attn128p<__bf16, 2>(a, s, thr); return;
            case 4: launch_attn128p<__bf16, 4>(a, s, thr); return;
            case 8: launch_attn128p<__bf16, 8>(a, s, thr); return;
            case 16: launch_attn128p<__bf16, 16>(a, s, thr); return;
            case 32: launch_attn128p<__bf16, 32>(a, s, thr); return;
            case 34: launch_attn128p<__bf16, 34>(a, s, thr); return;
            case 24: launch_attn128p<__bf16, 24>(a, s, thr); return;
            case 58: launch_attn128p<__bf16, 58>(a, s, thr); return;
            case 62: launch_attn128p<__bf16, 62>(a, s, thr); return;
            default: break;
        }
    }
#endif
    if (dt == DT_BF16) launch_attn128p<__bf16>(a, s, ov ? thr_override : ApT<__bf16>::thr);
    else launch_attn128p<_Float16>(a, s, ov ? fminf(thr_override, ApT<_Float16>::thr) : ApT<_Float16>::thr);
}

}  // namespace ldx
