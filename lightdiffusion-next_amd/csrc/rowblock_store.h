// Output stage shared by the row-block kernels (rowgemm.hip, xattn_block.hip, ff_block.hip).  Their MFMAs leave the results transposed — a lane
// holds one row and 4 consecutive features per accumulator tile — and storing those 8-byte pieces directly wrote every 128-byte line in four or
// more partial bursts from different waves: WRITE_SIZE was 2x the algorithmic bytes (profiles/r03/pmc_util_rowblock.txt) and the store phase
// 11 of xattn_block's 45 us.  Here the finished 16-bit values of 64 rows x 320 features go through an LDS tile and leave as whole 640-byte rows
// (16 bytes per lane, consecutive lanes = consecutive chunks).
#pragma once
#include "ldx_device.h"

namespace ldx {

constexpr int RB_SROW = 320 * 2 + 16;              // bytes per staged row
constexpr int RB_STAGE_BYTES = 64 * RB_SROW;       // 41 984

// out[t][qt]: the lane's packed values for row 16 qt + l15, features 40 wave + 16 t + 4 g4 .. + 3 (valid when 16 t + 4 g4 < 40) of a 320-wide pass;
// QT = 8 (128 rows, two halves) or 4 (64 rows).  Y + col0 = first feature of the pass; rows m0 .. m0 + 16 QT - 1 (clipped at M).
// dup_rows != 0: every row is also stored at row m + dup_rows.  Every thread of the 512-thread workgroup must call it (barriers inside); sS must not alias anything a slower wave may still read.
template <typename T, int QT>
__device__ __forceinline__ void rb_store_rows(T* __restrict__ Y, const int ldy, const long m0, const long M, const int col0, const int wave, const int l15,
                                              const int g4, const int tid, const uint2 (&out)[3][QT], char* sS, const long dup_rows = 0) {
    // the addresses below are invariant across the callers' pass / chunk loops; left to LICM they are all precomputed and kept live through the
    // MFMA loops (23 spilled dwords in rowgemm): make the lane ids opaque here so that they are recomputed where they are used
    int tid_ = tid, l15_ = l15;
    asm volatile("" : "+v"(tid_), "+v"(l15_));
#pragma unroll
    for (int hh = 0; hh < QT / 4; ++hh) {
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int nl = 16 * t + 4 * g4;
            if (nl < 40) {
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) *(uint2*)(sS + (16 * q4 + l15_) * RB_SROW + (wave * 40 + nl) * 2) = out[t][4 * hh + q4];
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int idx = tid_ + 512 * j, row = idx / 40, ch = idx - row * 40;
            const long m = m0 + 64 * hh + row;
            if (m < M) {
                const uint4 v = *(const uint4*)(sS + row * RB_SROW + ch * 16);
                *(uint4*)(Y + m * ldy + col0 + ch * 8) = v;
                if (dup_rows) *(uint4*)(Y + (m + dup_rows) * ldy + col0 + ch * 8) = v;       // second half of a shared CFG prefix (RowGemmArgs::dup_rows)
            }
        }
        __syncthreads();
    }
}

}  // namespace ldx
