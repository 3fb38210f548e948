// Boundary and scheduler-side elementwise kernels (fp32 at the edge, 16-bit NHWC inside).
//
// prep   : BaseModel.apply_model's input half (Model/ModelBase.py:72-112): EPS.calculate_input
//          (sample/sampling.py:29-35), ModelSamplingDiscrete.timestep (sampling.py:309-320) and the
//          timestep_embedding lookup (sample/sampling_util.py:56-76, table built by the host).
// finish : EPS.calculate_denoised (sampling.py:37-56).
// sampler_step / bilinear : the per-iteration latent update of samplers.sample_euler
//          (samplers.py:166-327) / sample_dpmpp_2m_cfgpp (samplers.py:754-962) incl. the CFG lerp
//          of CFG.cfg_function (CFG.py:55-60) and the multiscale F.interpolate(bilinear).
#include "ldx_device.h"
#include "ldx_kernels.h"

namespace ldx {

template <typename T>
static __device__ __forceinline__ void prep_image_body(const PrepArgs& p, const int block, const int nblocks) {
    // one thread per (b, pixel, 8-channel chunk) of the padded NHWC output
    const int HW = p.H * p.W;
    const int cpp = p.Cpad / 8;
    const long total = (long)p.B * HW * cpp;
    for (long idx = (long)block * 256 + threadIdx.x; idx < total; idx += (long)nblocks * 256) {
        const int ch = (int)(idx % cpp);
        const long bp = idx / cpp;
        const int pix = (int)(bp % HW), b = (int)(bp / HW);
        float scale = 1.f;
        if (p.scale_input) { const float sg = p.sigma[b]; scale = 1.0f / sqrtf(sg * sg + 1.0f); }
        const int bx = p.xB > 0 ? b % p.xB : b;
        const int Cx = p.cc ? p.Cx : p.C;
        float f[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = ch * 8 + e;
            f[e] = c < Cx ? p.x[((long)bx * Cx + c) * HW + pix] * scale : (c < p.C ? p.cc[((long)b * (p.C - Cx) + (c - Cx)) * HW + pix] : 0.f);
        }
        *(uint4*)((T*)p.xc + ((long)b * HW + pix) * p.Cpad + ch * 8) = pack8<T>(f);
    }
}

// ModelSamplingDiscrete.timestep (sample/sampling.py:309-320): index of the log-sigma table entry nearest to log(sigma), first minimum like
// torch.argmin.  One workgroup of 256 threads; sd / si are its scratch; every thread returns the index.
static __device__ __forceinline__ int nearest_log_sigma(const float sigma, const float* __restrict__ log_sigmas, const int n, float* sd, int* si) {
    const int tid = threadIdx.x;
    // log in fp64, rounded once: the correctly rounded fp32 logarithm.  The reference's torch CPU log is correctly rounded for 99.98 % of inputs (894 of 4 M
    // sampled differ), the device's logf for fewer — and at a near-tie (sigma at the geometric midpoint of two table entries) one ulp of log(sigma) decides the
    // index: with logf the golden sigma 0.36080566 (tests/golden/schedules.npz) came out as 108 instead of 107.  Host-side sigmas do not depend on this at all
    // (ldx_unet_denoise_cfg_t / ldx_unet_denoise_t carry the index computed by the reference's own expression).
    const float ls = (float)log((double)sigma);
    float best = INFINITY; int bi = 0x7fffffff;
    for (int k = tid; k < n; k += 256) {
        const float d = fabsf(ls - log_sigmas[k]);
        if (d < best) { best = d; bi = k; }
    }
    sd[tid] = best; si[tid] = bi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
            const float d2 = sd[tid + o]; const int i2 = si[tid + o];
            if (d2 < sd[tid] || (d2 == sd[tid] && i2 < si[tid])) { sd[tid] = d2; si[tid] = i2; }
        }
        __syncthreads();
    }
    const int t = si[0];
    __syncthreads();
    return t;
}
// one block per (sample, slice): timestep index (given, or looked up from sigma), then copy the host-built sinusoidal embedding row.
static __device__ __forceinline__ void prep_time_body(const PrepArgs& p, const int b, const int part, const int nparts) {
    __shared__ float sd[256];
    __shared__ int si[256];
    const int tid = threadIdx.x;
    int t = p.t_in ? (int)p.t_in[b] : nearest_log_sigma(p.sigma[b], p.log_sigmas, p.n_sigmas, sd, si);
    t = max(0, min(t, p.n_sigmas - 1));
    // nparts slices per sample: every slice finds the same index, slice 0 writes the embedding row and the timestep, all of them share the emb_layers row copy
    // (one workgroup per sample walked its 70 KB in 17 dependent 4-KiB rounds: 14 us of a 14 ms step)
    if (part == 0)
        for (int j = tid; j < p.temb_dim; j += 256) p.temb_out[(long)b * p.temb_dim + j] = p.temb_table[(long)t * p.temb_dim + j];
    if (p.emb_table) {                                   // emb_n % 4 == 0 (channel counts are multiples of 64)
        const float4* src = (const float4*)(p.emb_table + (long)t * p.emb_n);
        float4* dst = (float4*)(p.emb_out + (long)b * p.emb_n);
        for (int j = tid + 256 * part; j < p.emb_n / 4; j += 256 * nparts) dst[j] = src[j];
    }
    if (part == 0 && tid == 0 && p.t_out) p.t_out[b] = (float)t;
}
// the lookup alone (ldx_unet_timestep): the SAME device function the prep kernel runs, exposed so that the index can be tested as an integer
__global__ __launch_bounds__(256) void timestep_kernel(const float* __restrict__ sigma, const float* __restrict__ log_sigmas, const int n_sigmas, int* __restrict__ out) {
    __shared__ float sd[256];
    __shared__ int si[256];
    const int t = nearest_log_sigma(sigma[blockIdx.x], log_sigmas, n_sigmas, sd, si);
    if (threadIdx.x == 0) out[blockIdx.x] = max(0, min(t, n_sigmas - 1));
}
void launch_timestep(const float* sigma, const float* log_sigmas, int n_sigmas, int n, int* out, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(timestep_kernel, dim3(n), dim3(256), 0, s, sigma, log_sigmas, n_sigmas, out);
}

// ONE launch for both halves of the boundary (round 6; they were two): blocks [0, gimg) convert the image, the next B * nparts blocks do the timestep part
// (the two do not depend on each other).  The kernel keeps the name prep_image_kernel: profiles/analyze_trace.py finds the start of a forward by it.
template <typename T>
__global__ __launch_bounds__(256) void prep_image_kernel(const PrepArgs p, const int gimg, const int nparts) {
    if ((int)blockIdx.x < gimg) prep_image_body<T>(p, blockIdx.x, gimg);
    else { const int t = (int)blockIdx.x - gimg; prep_time_body(p, t / nparts, t % nparts, nparts); }
}
void launch_prep(const PrepArgs& a, DType dt, hipStream_t s) {
    const long total = (long)a.B * a.H * a.W * (a.Cpad / 8);
    int grid = (int)((total + 255) / 256); if (grid > 4096) grid = 4096; if (grid < 1) grid = 1;
    const int nparts = a.emb_table ? 16 : 1, gt = a.temb_out ? a.B * nparts : 0;
    if (dt == DT_BF16) hipLaunchKernelGGL((prep_image_kernel<__bf16>), dim3(grid + gt), dim3(256), 0, s, a, grid, nparts);
    else hipLaunchKernelGGL((prep_image_kernel<_Float16>), dim3(grid + gt), dim3(256), 0, s, a, grid, nparts);
}

__global__ __launch_bounds__(256) void finish_kernel(const FinishArgs p) {
    const long total = (long)p.B * p.C * p.HW;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int pix = (int)(idx % p.HW);
        const long bc = idx / p.HW;
        const int c = (int)(bc % p.C), b = (int)(bc / p.C);
        const float e = p.eps[((long)b * p.HW + pix) * p.ld + c];
        const long xi = p.xB > 0 ? ((long)(b % p.xB) * p.C + c) * p.HW + pix : idx;
        p.out[idx] = p.x ? (p.x[xi] - e * p.sigma[b]) : e;
    }
}
// CLIPTextModel_.forward's pooled_output (clip/CLIPTextModel.py:98-106) and CLIPTextModel.forward's text_projection (:152-163): one block per sample
__global__ __launch_bounds__(256) void clip_pooled_kernel(const float* last, const int* ids, int T, int E, int eos_id, const float* proj, float* out) {
    extern __shared__ float srow[];                  // [E]
    __shared__ int spos;
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) {
        int pos = 0;
        for (int t = 0; t < T; ++t) if (ids[(long)b * T + t] == eos_id) { pos = t; break; }
        spos = pos;
    }
    __syncthreads();
    const float* row = last + ((long)b * T + spos) * E;
    for (int k = tid; k < E; k += 256) srow[k] = row[k];
    __syncthreads();
    for (int n = tid; n < E; n += 256) {
        float acc;
        if (proj) {
            acc = 0.f;
            const float* w = proj + (long)n * E;
            for (int k = 0; k < E; ++k) acc = fmaf(srow[k], w[k], acc);
        } else acc = srow[n];
        out[(long)b * E + n] = acc;
    }
}
void launch_clip_pooled(const float* last, const int* ids, int B, int T, int E, int eos_id, const float* proj, float* out, hipStream_t s) {
    hipLaunchKernelGGL(clip_pooled_kernel, dim3(B), dim3(256), E * sizeof(float), s, last, ids, T, E, eos_id, proj, out);
}
__global__ void fill_f32_kernel(float* dst, float v, int n) { const int i = blockIdx.x * 64 + threadIdx.x; if (i < n) dst[i] = v; }
__global__ void fill2_f32_kernel(float* a, float va, float* b, float vb, int n) { const int i = blockIdx.x * 64 + threadIdx.x; if (i < n) { a[i] = va; b[i] = vb; } }
__global__ __launch_bounds__(256) void dup_rows_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, const long total, const int cpr, const int ld16) {
    // 4 chunks in flight per thread (a copy is pure latency: 10 MB at 1024^2)
    const long stride = (long)gridDim.x * 256;
    for (long i0 = (long)blockIdx.x * 256 + threadIdx.x; i0 < total; i0 += 4 * stride) {
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const long i = i0 + u * stride; if (i < total) { const long r = i / cpr; v[u] = src[r * ld16 + (i - r * cpr)]; } }
#pragma unroll
        for (int u = 0; u < 4; ++u) { const long i = i0 + u * stride; if (i < total) { const long r = i / cpr; dst[r * ld16 + (i - r * cpr)] = v[u]; } }
    }
}
void launch_dup_rows(const void* src, void* dst, int rows, int C, int ld, DType, hipStream_t s) {
    if (rows <= 0 || C <= 0) return;
    const long total = (long)rows * (C / 8);
    long grid = (total + 256 * 4 - 1) / (256 * 4); if (grid > 2048) grid = 2048; if (grid < 1) grid = 1;
    hipLaunchKernelGGL(dup_rows_kernel, dim3((unsigned)grid), dim3(256), 0, s, (const uint4*)src, (uint4*)dst, total, C / 8, ld / 8);
}
void launch_fill2_f32(float* a, float va, float* b, float vb, int n, hipStream_t s) { hipLaunchKernelGGL(fill2_f32_kernel, dim3((n + 63) / 64), dim3(64), 0, s, a, va, b, vb, n); }
void launch_fill_f32(float* dst, float v, int n, hipStream_t s) { hipLaunchKernelGGL(fill_f32_kernel, dim3((n + 63) / 64), dim3(64), 0, s, dst, v, n); }
void launch_finish(const FinishArgs& a, hipStream_t s) {
    const long total = (long)a.B * a.C * a.HW;
    int grid = (int)((total + 255) / 256); if (grid > 2048) grid = 2048; if (grid < 1) grid = 1;
    hipLaunchKernelGGL(finish_kernel, dim3(grid), dim3(256), 0, s, a);
}

template <typename T>
__global__ void f32_to_t_kernel(const float* in, T* out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = (T)in[i];
}
template <typename T>
__global__ void t_to_f32_kernel(const T* in, float* out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = (float)in[i];
}
static int grid_for(size_t n) { size_t g = (n + 255) / 256; if (g > 8192) g = 8192; if (g < 1) g = 1; return (int)g; }
void launch_f32_to_t(const float* in, void* out, size_t n, DType dt, hipStream_t s) {
    if (dt == DT_BF16) hipLaunchKernelGGL((f32_to_t_kernel<__bf16>), dim3(grid_for(n)), dim3(256), 0, s, in, (__bf16*)out, n);
    else hipLaunchKernelGGL((f32_to_t_kernel<_Float16>), dim3(grid_for(n)), dim3(256), 0, s, in, (_Float16*)out, n);
}
void launch_t_to_f32(const void* in, float* out, size_t n, DType dt, hipStream_t s) {
    if (dt == DT_BF16) hipLaunchKernelGGL((t_to_f32_kernel<__bf16>), dim3(grid_for(n)), dim3(256), 0, s, (const __bf16*)in, out, n);
    else hipLaunchKernelGGL((t_to_f32_kernel<_Float16>), dim3(grid_for(n)), dim3(256), 0, s, (const _Float16*)in, out, n);
}

template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* in, T* out, int B, int C, int HW, int Cpad, float scale) {
    const int cpp = Cpad / 8;
    const long total = (long)B * HW * cpp;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int ch = (int)(idx % cpp);
        const long bp = idx / cpp;
        const int pix = (int)(bp % HW), b = (int)(bp / HW);
        float f[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = ch * 8 + e;
            f[e] = (c < C) ? in[((long)b * C + c) * HW + pix] * scale : 0.f;
        }
        *(uint4*)(out + ((long)b * HW + pix) * Cpad + ch * 8) = pack8<T>(f);
    }
}
void launch_nchw_to_nhwc(const float* in, void* out, int B, int C, int HW, int Cpad, float scale, DType dt, hipStream_t s) {
    const size_t total = (size_t)B * HW * (Cpad / 8);
    if (dt == DT_BF16) hipLaunchKernelGGL((nchw_to_nhwc_kernel<__bf16>), dim3(grid_for(total)), dim3(256), 0, s, in, (__bf16*)out, B, C, HW, Cpad, scale);
    else hipLaunchKernelGGL((nchw_to_nhwc_kernel<_Float16>), dim3(grid_for(total)), dim3(256), 0, s, in, (_Float16*)out, B, C, HW, Cpad, scale);
}

template <typename T>
__global__ __launch_bounds__(256) void vae_prep_kernel(const float* z, T* out, int B, int C, int HW, int Cpad, const float* mw, const float* mb) {
    const int cpp = Cpad / 8;
    const long total = (long)B * HW * cpp;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int ch = (int)(idx % cpp);
        const long bp = idx / cpp;
        const int pix = (int)(bp % HW), b = (int)(bp / HW);
        float f[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = ch * 8 + e;
            float v = 0.f;
            if (c < C) {
                if (mw) {
                    v = mb ? mb[c] : 0.f;
                    for (int k = 0; k < C; ++k) v += mw[c * C + k] * z[((long)b * C + k) * HW + pix];
                } else v = z[((long)b * C + c) * HW + pix];
            }
            f[e] = v;
        }
        *(uint4*)(out + ((long)b * HW + pix) * Cpad + ch * 8) = pack8<T>(f);
    }
}
void launch_vae_prep(const float* z, void* out, int B, int C, int HW, int Cpad, const float* mw, const float* mb, DType dt, hipStream_t s) {
    const size_t total = (size_t)B * HW * (Cpad / 8);
    if (dt == DT_BF16) hipLaunchKernelGGL((vae_prep_kernel<__bf16>), dim3(grid_for(total)), dim3(256), 0, s, z, (__bf16*)out, B, C, HW, Cpad, mw, mb);
    else hipLaunchKernelGGL((vae_prep_kernel<_Float16>), dim3(grid_for(total)), dim3(256), 0, s, z, (_Float16*)out, B, C, HW, Cpad, mw, mb);
}

template <typename T>
__global__ __launch_bounds__(256) void pixels_prep_kernel(const float* px, T* out, int B, int C, int HW, int Cpad, float scale, float shift) {
    const int cpp = Cpad / 8;
    const long total = (long)B * HW * cpp;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int ch = (int)(idx % cpp);
        const long bp = idx / cpp;
        float f[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { const int c = ch * 8 + e; f[e] = c < C ? px[bp * C + c] * scale + shift : 0.f; }
        *(uint4*)(out + bp * Cpad + ch * 8) = pack8<T>(f);
    }
}
void launch_pixels_prep(const float* px, void* out, int B, int C, int HW, int Cpad, float scale, float shift, DType dt, hipStream_t s) {
    const size_t total = (size_t)B * HW * (Cpad / 8);
    if (dt == DT_BF16) hipLaunchKernelGGL((pixels_prep_kernel<__bf16>), dim3(grid_for(total)), dim3(256), 0, s, px, (__bf16*)out, B, C, HW, Cpad, scale, shift);
    else hipLaunchKernelGGL((pixels_prep_kernel<_Float16>), dim3(grid_for(total)), dim3(256), 0, s, px, (_Float16*)out, B, C, HW, Cpad, scale, shift);
}

__global__ __launch_bounds__(256) void clamp01_kernel(const float* in, float* out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        out[i] = fminf(fmaxf((in[i] + 1.0f) / 2.0f, 0.0f), 1.0f);
}
void launch_clamp01(const float* in, float* out, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(clamp01_kernel, dim3(grid_for(n)), dim3(256), 0, s, in, out, n);
}

// one workgroup per row; three passes over the (L2-resident) row: max, sum of exp, normalise
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_kernel(T* X, int cols, int ld, float scale) {
    __shared__ float red[4];
    T* __restrict__ x = X + (long)blockIdx.x * ld;
    const int tid = threadIdx.x, nch = cols >> 3;
    const float c = scale * 1.44269504088896340736f;
    float mx = -INFINITY;
    for (int ch = tid; ch < nch; ch += 256) {
        float f[8]; unpack8<T>(*(const uint4*)(x + ch * 8), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) mx = fmaxf(mx, f[e]);
    }
    mx = wave_max(mx);
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
    for (int ch = tid; ch < nch; ch += 256) {
        float f[8]; unpack8<T>(*(const uint4*)(x + ch * 8), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) sum += __builtin_amdgcn_exp2f((f[e] - mx) * c);
    }
    sum = wave_sum(sum);
    if ((tid & 63) == 0) red[tid >> 6] = sum;
    __syncthreads();
    const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
    for (int ch = tid; ch < nch; ch += 256) {
        float f[8]; unpack8<T>(*(const uint4*)(x + ch * 8), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = __builtin_amdgcn_exp2f((f[e] - mx) * c) * inv;
        *(uint4*)(x + ch * 8) = pack8<T>(f);
    }
}
void launch_softmax_rows(void* X, int rows, int cols, int ld, float scale, DType dt, hipStream_t s) {
    if (dt == DT_BF16) hipLaunchKernelGGL((softmax_rows_kernel<__bf16>), dim3(rows), dim3(256), 0, s, (__bf16*)X, cols, ld, scale);
    else hipLaunchKernelGGL((softmax_rows_kernel<_Float16>), dim3(rows), dim3(256), 0, s, (_Float16*)X, cols, ld, scale);
}

template <typename T>
__global__ __launch_bounds__(256) void clip_embed_kernel(const int* ids, const float* tok, const float* pos, T* out, int B, int Tn, int C, int vocab,
                                                          const float* extra, int n_extra) {
    const long total = (long)B * Tn * (C / 8);
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int ch = (int)(idx % (C / 8));
        const long bt = idx / (C / 8);
        const int t = (int)(bt % Tn);
        int id = ids[bt];
        id = max(0, min(id, vocab + n_extra - 1));
        // ids >= vocab address the textual-inversion rows appended to the table for this call (SDClip.py:213-267)
        const float* row = id < vocab ? tok + (long)id * C : extra + (long)(id - vocab) * C;
        float f[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = row[ch * 8 + e] + (pos ? pos[(long)t * C + ch * 8 + e] : 0.f);
        *(uint4*)(out + bt * C + ch * 8) = pack8<T>(f);
    }
}
void launch_clip_embed(const int* ids, const float* tok, const float* pos, void* out, int B, int Tn, int C, int vocab, const float* extra, int n_extra, DType dt, hipStream_t s) {
    const size_t total = (size_t)B * Tn * (C / 8);
    if (dt == DT_BF16) hipLaunchKernelGGL((clip_embed_kernel<__bf16>), dim3(grid_for(total)), dim3(256), 0, s, ids, tok, pos, (__bf16*)out, B, Tn, C, vocab, extra, n_extra);
    else hipLaunchKernelGGL((clip_embed_kernel<_Float16>), dim3(grid_for(total)), dim3(256), 0, s, ids, tok, pos, (_Float16*)out, B, Tn, C, vocab, extra, n_extra);
}

__global__ __launch_bounds__(256) void flux_temb_kernel(const float* t, float* out, int B, int dim, float factor) {
    const int half = dim / 2;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < B * half; idx += gridDim.x * 256) {
        const int b = idx / half, j = idx % half;
        const float freq = expf(-9.210340371976184f * (float)j / (float)half);      // exp(-ln(10000) j / half)
        const float a = (factor * t[b]) * freq;
        out[(long)b * dim + j] = cosf(a);
        out[(long)b * dim + half + j] = sinf(a);
    }
}
void launch_flux_temb(const float* t, float* out, int B, int dim, float factor, hipStream_t s) {
    hipLaunchKernelGGL(flux_temb_kernel, dim3(grid_for((size_t)B * dim / 2)), dim3(256), 0, s, t, out, B, dim, factor);
}
__global__ __launch_bounds__(256) void silu_f32_kernel(const float* in, float* out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const float x = in[i]; out[i] = x / (1.0f + expf(-x)); }
}
void launch_silu_f32(const float* in, float* out, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(silu_f32_kernel, dim3(grid_for(n)), dim3(256), 0, s, in, out, n);
}
template <typename T>
__global__ __launch_bounds__(256) void flux_patchify_kernel(const float* x, T* out, int B, int C, int H, int W) {
    const int h2 = H / 2, w2 = W / 2, K = 4 * C;
    const long total = (long)B * h2 * w2 * K;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int col = (int)(idx % K);
        const long tokb = idx / K;
        const int tw = (int)(tokb % w2), th = (int)((tokb / w2) % h2), b = (int)(tokb / ((long)w2 * h2));
        const int c = col >> 2, ph = (col >> 1) & 1, pw = col & 1;
        out[idx] = (T)x[(((long)b * C + c) * H + th * 2 + ph) * W + tw * 2 + pw];
    }
}
void launch_flux_patchify(const float* x, void* out, int B, int C, int H, int W, DType dt, hipStream_t s) {
    const size_t total = (size_t)B * (H / 2) * (W / 2) * 4 * C;
    if (dt == DT_BF16) hipLaunchKernelGGL((flux_patchify_kernel<__bf16>), dim3(grid_for(total)), dim3(256), 0, s, x, (__bf16*)out, B, C, H, W);
    else hipLaunchKernelGGL((flux_patchify_kernel<_Float16>), dim3(grid_for(total)), dim3(256), 0, s, x, (_Float16*)out, B, C, H, W);
}
__global__ __launch_bounds__(256) void flux_unpatchify_kernel(const float* tok, int ld, const float* x, const float* sigma, float* out, int B, int C, int H, int W) {
    const int h2 = H / 2, w2 = W / 2;
    const long total = (long)B * C * H * W;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int xw = (int)(idx % W), yh = (int)((idx / W) % H), c = (int)((idx / ((long)W * H)) % C), b = (int)(idx / ((long)W * H * C));
        const long t = ((long)b * h2 + (yh >> 1)) * w2 + (xw >> 1);
        const float v = tok[t * ld + c * 4 + (yh & 1) * 2 + (xw & 1)];
        out[idx] = x ? (x[idx] - v * sigma[b]) : v;
    }
}
void launch_flux_unpatchify(const float* tok, int ld, const float* x, const float* sigma, float* out, int B, int C, int H, int W, hipStream_t s) {
    hipLaunchKernelGGL(flux_unpatchify_kernel, dim3(grid_for((size_t)B * C * H * W)), dim3(256), 0, s, tok, ld, x, sigma, out, B, C, H, W);
}

#pragma clang fp contract(off)
__global__ __launch_bounds__(256) void sampler_step_kernel(const StepArgs p) {
    // Same operation order (and no FMA contraction) as the reference's fp32 tensor expressions.
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < p.n; i += (size_t)gridDim.x * 256) {
        const float u = p.den_uncond[i], c = p.den_cond[i];
        // torch.lerp(u, c, w): |w| < 0.5 ? u + w*(c-u) : c - (c-u)*(1-w)   (ATen Lerp.h)
        const float diff = c - u;
        const float d = (fabsf(p.cfg) < 0.5f) ? (u + p.cfg * diff) : (c - diff * (1.0f - p.cfg));
        if (p.denoised_out) p.denoised_out[i] = d;
        if (p.kind == 2) continue;                           // CFG combine only
        if (p.kind == 3) { p.x[i] = p.x[i] + u * p.c0; continue; }   // noise injection (den_uncond = noise)
        const float x = p.x[i];
        float xn;
        if (p.kind == 0) xn = x + ((x - d) / p.c0) * p.c1;   // c0 = sigma_hat, c1 = sigma_next - sigma_hat
        else xn = p.c0 * x - p.c1 * d;                       // c0 = sigma_next/sigma, c1 = expm1(-h)
        p.x[i] = xn;
    }
}
#pragma clang fp contract(fast)
void launch_sampler_step(const StepArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(sampler_step_kernel, dim3(grid_for(a.n)), dim3(256), 0, s, a);
}

// slerp of upscale.py:8-40, one output pixel per thread (C <= 16 channels)
__global__ __launch_bounds__(256) void bislerp_kernel(const float* in, float* out, int N, int C, int H, int W, int axis, int L,
                                                      const int* c1, const int* c2, const float* rt) {
    const int Ho = axis == 0 ? L : H, Wo = axis == 1 ? L : W;
    const long total = (long)N * Ho * Wo;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int xo = (int)(idx % Wo), yo = (int)((idx / Wo) % Ho), n = (int)(idx / ((long)Wo * Ho));
        const int i = axis == 1 ? xo : yo;
        const float r = rt[i];
        const int y1 = axis == 0 ? c1[i] : yo, y2 = axis == 0 ? c2[i] : yo, x1 = axis == 1 ? c1[i] : xo, x2 = axis == 1 ? c2[i] : xo;
        float b1[16], b2[16];
        float n1 = 0.f, n2 = 0.f;
        for (int c = 0; c < C; ++c) {
            b1[c] = in[(((long)n * C + c) * H + y1) * W + x1];
            b2[c] = in[(((long)n * C + c) * H + y2) * W + x2];
            n1 += b1[c] * b1[c]; n2 += b2[c] * b2[c];
        }
        n1 = sqrtf(n1); n2 = sqrtf(n2);
        float dot = 0.f;
        for (int c = 0; c < C; ++c) {
            const float u = n1 == 0.f ? 0.f : b1[c] / n1, v = n2 == 0.f ? 0.f : b2[c] / n2;
            dot += u * v;
        }
        const float omega = acosf(dot), so = sinf(omega);
        const float w1 = sinf((1.0f - r) * omega) / so, w2 = sinf(r * omega) / so;
        const float nm = n1 * (1.0f - r) + n2 * r;
        for (int c = 0; c < C; ++c) {
            const float u = n1 == 0.f ? 0.f : b1[c] / n1, v = n2 == 0.f ? 0.f : b2[c] / n2;
            float res = (w1 * u + w2 * v) * nm;
            if (dot > 1.0f - 1e-5f) res = b1[c];
            if (dot < 1e-5f - 1.0f) res = b1[c] * (1.0f - r) + b2[c] * r;
            out[(((long)n * C + c) * Ho + yo) * Wo + xo] = res;
        }
    }
}
void launch_bislerp_pass(const float* in, float* out, int N, int C, int H, int W, int axis, int new_len,
                         const int* c1, const int* c2, const float* r, hipStream_t s) {
    const size_t total = (size_t)N * (axis == 0 ? new_len : H) * (axis == 1 ? new_len : W);
    hipLaunchKernelGGL(bislerp_kernel, dim3(grid_for(total)), dim3(256), 0, s, in, out, N, C, H, W, axis, new_len, c1, c2, r);
}

__global__ __launch_bounds__(256) void mix_nhwc_to_nchw_kernel(const float* in, int ld, float* out, int B, int C, int HW, const float* w, const float* bias) {
    const long total = (long)B * C * HW;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int p = (int)(idx % HW), c = (int)((idx / HW) % C), b = (int)(idx / ((long)HW * C));
        const float* src = in + ((long)b * HW + p) * ld;
        float v = bias ? bias[c] : 0.f;
        if (w) { for (int k = 0; k < C; ++k) v += w[c * C + k] * src[k]; } else v += src[c];
        out[idx] = v;
    }
}
void launch_mix_nhwc_to_nchw(const float* in, int ld, float* out, int B, int C, int HW, const float* w, const float* bias, hipStream_t s) {
    hipLaunchKernelGGL(mix_nhwc_to_nchw_kernel, dim3(grid_for((size_t)B * C * HW)), dim3(256), 0, s, in, ld, out, B, C, HW, w, bias);
}

// torch upsample_bilinear2d, align_corners=False: src = max((dst + .5) * in/out - .5, 0)
__global__ __launch_bounds__(256) void bilinear_kernel(const float* in, float* out, int planes, int Hin, int Win, int Hout, int Wout) {
    const float sy = (float)Hin / (float)Hout, sx = (float)Win / (float)Wout;
    const long total = (long)planes * Hout * Wout;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int ox = (int)(idx % Wout);
        const long r = idx / Wout;
        const int oy = (int)(r % Hout), pl = (int)(r / Hout);
        const float fy = fmaxf(((float)oy + 0.5f) * sy - 0.5f, 0.f);
        const float fx = fmaxf(((float)ox + 0.5f) * sx - 0.5f, 0.f);
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = min(y0 + 1, Hin - 1), x1 = min(x0 + 1, Win - 1);
        const float ly = fy - (float)y0, lx = fx - (float)x0;
        const float* pin = in + (long)pl * Hin * Win;
        const float v00 = pin[y0 * Win + x0], v01 = pin[y0 * Win + x1];
        const float v10 = pin[y1 * Win + x0], v11 = pin[y1 * Win + x1];
        out[idx] = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
    }
}
void launch_bilinear(const float* in, float* out, int planes, int Hin, int Win, int Hout, int Wout, hipStream_t s) {
    const size_t total = (size_t)planes * Hout * Wout;
    hipLaunchKernelGGL(bilinear_kernel, dim3(grid_for(total)), dim3(256), 0, s, in, out, planes, Hin, Win, Hout, Wout);
}

// ------------------------------------------------------------------------------------------
// First-block cache (WaveSpeed/first_block_cache.py:105-148): elementwise helpers, HBM-bound, 8 values per thread-iteration.
template <typename T, int MODE>     // MODE 0: diff sums, 1: store first residual
__global__ __launch_bounds__(256) void fb_img_kernel(const T* X, const T* S0, float* F, int B, int L, int Lt, int C, float* partial) {
    const int Li = L - Lt, cpr = C / 8;
    const long total = (long)B * Li * cpr;
    float sd = 0.f, sp = 0.f;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int ch = (int)(idx % cpr);
        const long r = idx / cpr;
        const int b = (int)(r / Li), row = (int)(r % Li);
        const long xo = ((long)b * L + Lt + row) * C + ch * 8, fo = ((long)b * Li + row) * C + ch * 8;
        float x[8], s0[8];
        unpack8<T>(*(const uint4*)(X + xo), x);
        unpack8<T>(*(const uint4*)(S0 + xo), s0);
        if (MODE == 0) {
            const float4 f0 = *(const float4*)(F + fo), f1 = *(const float4*)(F + fo + 4);
            const float f[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) { sd += fabsf(f[e] - (x[e] - s0[e])); sp += fabsf(f[e]); }
        } else {
            *(float4*)(F + fo) = make_float4(x[0] - s0[0], x[1] - s0[1], x[2] - s0[2], x[3] - s0[3]);
            *(float4*)(F + fo + 4) = make_float4(x[4] - s0[4], x[5] - s0[5], x[6] - s0[6], x[7] - s0[7]);
        }
    }
    if (MODE == 0) {
        __shared__ float red[2][4];
        sd = wave_sum(sd); sp = wave_sum(sp);
        if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = sd; red[1][threadIdx.x >> 6] = sp; }
        __syncthreads();
        if (threadIdx.x == 0) {
            partial[blockIdx.x] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
            partial[gridDim.x + blockIdx.x] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
        }
    }
}
__global__ __launch_bounds__(256) void fb_reduce_kernel(const float* partial, int n, float* sums) {
    __shared__ float red[2][4];
    float a = 0.f, b = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) { a += partial[i]; b += partial[n + i]; }      // fixed order per thread
    a = wave_sum(a); b = wave_sum(b);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) { sums[0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]); sums[1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]); }
}
void launch_fb_diff(const void* X, const void* S0, const float* F, int B, int L, int Lt, int C, float* partial, float* sums, DType dt, hipStream_t s) {
    const size_t total = (size_t)B * (L - Lt) * (C / 8);
    int grid = (int)((total + 255) / 256); if (grid > 1024) grid = 1024; if (grid < 1) grid = 1;
    if (dt == DT_BF16) hipLaunchKernelGGL((fb_img_kernel<__bf16, 0>), dim3(grid), dim3(256), 0, s, (const __bf16*)X, (const __bf16*)S0, (float*)F, B, L, Lt, C, partial);
    else hipLaunchKernelGGL((fb_img_kernel<_Float16, 0>), dim3(grid), dim3(256), 0, s, (const _Float16*)X, (const _Float16*)S0, (float*)F, B, L, Lt, C, partial);
    hipLaunchKernelGGL(fb_reduce_kernel, dim3(1), dim3(256), 0, s, partial, grid, sums);
}
void launch_fb_first(const void* X, const void* S0, float* F, int B, int L, int Lt, int C, DType dt, hipStream_t s) {
    const size_t total = (size_t)B * (L - Lt) * (C / 8);
    if (dt == DT_BF16) hipLaunchKernelGGL((fb_img_kernel<__bf16, 1>), dim3(grid_for(total)), dim3(256), 0, s, (const __bf16*)X, (const __bf16*)S0, F, B, L, Lt, C, nullptr);
    else hipLaunchKernelGGL((fb_img_kernel<_Float16, 1>), dim3(grid_for(total)), dim3(256), 0, s, (const _Float16*)X, (const _Float16*)S0, F, B, L, Lt, C, nullptr);
}
template <typename T, int MODE>     // MODE 0: R = X - S1 ; 1: X += R
__global__ __launch_bounds__(256) void fb_joint_kernel(T* X, const T* S1, float* R, size_t n8) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
        float x[8];
        unpack8<T>(*(const uint4*)(X + i * 8), x);
        if (MODE == 0) {
            float s1[8];
            unpack8<T>(*(const uint4*)(S1 + i * 8), s1);
            *(float4*)(R + i * 8) = make_float4(x[0] - s1[0], x[1] - s1[1], x[2] - s1[2], x[3] - s1[3]);
            *(float4*)(R + i * 8 + 4) = make_float4(x[4] - s1[4], x[5] - s1[5], x[6] - s1[6], x[7] - s1[7]);
        } else {
            const float4 r0 = *(const float4*)(R + i * 8), r1 = *(const float4*)(R + i * 8 + 4);
            x[0] += r0.x; x[1] += r0.y; x[2] += r0.z; x[3] += r0.w; x[4] += r1.x; x[5] += r1.y; x[6] += r1.z; x[7] += r1.w;
            *(uint4*)(X + i * 8) = pack8<T>(x);
        }
    }
}
void launch_fb_residual(const void* X, const void* S1, float* R, size_t n, DType dt, hipStream_t s) {
    if (dt == DT_BF16) hipLaunchKernelGGL((fb_joint_kernel<__bf16, 0>), dim3(grid_for(n / 8)), dim3(256), 0, s, (__bf16*)X, (const __bf16*)S1, R, n / 8);
    else hipLaunchKernelGGL((fb_joint_kernel<_Float16, 0>), dim3(grid_for(n / 8)), dim3(256), 0, s, (_Float16*)X, (const _Float16*)S1, R, n / 8);
}
void launch_fb_apply(void* X, const float* R, size_t n, DType dt, hipStream_t s) {
    if (dt == DT_BF16) hipLaunchKernelGGL((fb_joint_kernel<__bf16, 1>), dim3(grid_for(n / 8)), dim3(256), 0, s, (__bf16*)X, nullptr, (float*)R, n / 8);
    else hipLaunchKernelGGL((fb_joint_kernel<_Float16, 1>), dim3(grid_for(n / 8)), dim3(256), 0, s, (_Float16*)X, nullptr, (float*)R, n / 8);
}

// ------------------------------------------------------------------------------------------
// tiled_scale blending (Utilities/util.py:557-590)
__global__ __launch_bounds__(256) void tile_blend_kernel(const float* tile, int th, int tw, float* out, float* div, int H, int W, int C,
                                                         int y0, int x0, int feather) {
    const long total = (long)th * tw * C;
    const bool fy = feather < th, fx = feather < tw;                  // "if feather >= mask.shape[d]: continue"
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = (int)(idx % C);
        const long px = idx / C;
        const int x = (int)(px % tw), y = (int)(px / tw);
        float m = 1.0f;
        // mask.narrow(d, t, 1).mul_(a) and .narrow(d, size-1-t, 1).mul_(a), a = (t+1)/feather: both may hit the same line
        if (fy) { if (y < feather) m *= (float)(y + 1) / (float)feather; if (th - 1 - y < feather) m *= (float)(th - y) / (float)feather; }
        if (fx) { if (x < feather) m *= (float)(x + 1) / (float)feather; if (tw - 1 - x < feather) m *= (float)(tw - x) / (float)feather; }
        const long o = ((long)(y0 + y) * W + (x0 + x)) * C + c;
        out[o] += tile[idx] * m;
        div[o] += m;
    }
}
void launch_tile_blend(const float* tile, int th, int tw, float* out, float* div, int H, int W, int C, int y0, int x0, int feather, hipStream_t s) {
    hipLaunchKernelGGL(tile_blend_kernel, dim3(grid_for((size_t)th * tw * C)), dim3(256), 0, s, tile, th, tw, out, div, H, W, C, y0, x0, feather);
}
__global__ __launch_bounds__(256) void tile_finish_kernel(float* out, const float* div, size_t n, int clamp01) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float v = div ? out[i] / div[i] : out[i];
        if (clamp01) v = fminf(fmaxf(v, 0.f), 1.f);
        out[i] = v;
    }
}
void launch_tile_finish(float* out, const float* div, size_t n, int clamp01, hipStream_t s) {
    hipLaunchKernelGGL(tile_finish_kernel, dim3(grid_for(n)), dim3(256), 0, s, out, div, n, clamp01);
}

}  // namespace ldx
