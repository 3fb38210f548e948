// Where does a 256-row ping-pong tile spend its life?  The product kernel (gemm_pp.inc, included as is) with wall-clock stamps per workgroup:
//   0 kernel entry, 1 descriptors / offsets done (first DMA issue), 2 first operands landed (first barrier), 3 main loop done, 5 per-column operands staged in LDS,
//   6 every store of wave 0 issued, 4 output stage done (stores retired).
// Usage: pp_stamp [mx|bf16] M N K BN [resid|geglu|lngeglu] ; prints per-phase averages over the workgroups and the launch's span.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -pragma-unroll-threshold=1000000 -I lightdiffusion-next_amd/csrc profiles/ubench/pp_stamp.hip -o profiles/ubench/pp_stamp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

__device__ unsigned long long g_stamp[8 * 8192];
#define LDX_PP_STAMP(k) do { if ((k) == 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); \
                             if (threadIdx.x == 0) g_stamp[blockIdx.x * 8 + (k)] = wall_clock64(); } while (0)
#include "gemm_common.h"
namespace ldx {
#include "gemm_pp.inc"
}
using namespace ldx;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <bool F8, int BN, bool LNF = false>
static void run(GemmArgs a, int reps) {
    const int tiles = ((a.M + 255) / 256) * ((a.N + BN - 1) / BN);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch_pp_inst<__bf16, 0, BN, LNF, F8>(a, 1, 0);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) launch_pp_inst<__bf16, 0, BN, LNF, F8>(a, 1, 0);
    CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> st(8 * 8192);
    CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_stamp), st.size() * 8));
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int b = 0; b < tiles; ++b) { t0 = std::min(t0, st[b * 8]); t1 = std::max(t1, st[b * 8 + 4]); }
    double ph[4] = {0, 0, 0, 0}, start = 0, startmax = 0, e5 = 0, e6 = 0;
    for (int b = 0; b < tiles; ++b) {
        for (int k = 0; k < 4; ++k) ph[k] += (double)(st[b * 8 + k + 1] - st[b * 8 + k]);
        e5 += (double)(st[b * 8 + 5] - st[b * 8 + 3]); e6 += (double)(st[b * 8 + 6] - st[b * 8 + 5]);
        start += (double)(st[b * 8] - t0); startmax = std::max(startmax, (double)(st[b * 8] - t0));
    }
    const double tick = 0.01;       // wall_clock64: 100 MHz
    printf("%s M%d N%d K%d BN%d %s: %d tiles, %.1f us per launch (events, back to back) | span first entry -> last store %.1f us | per workgroup avg: entry after first +%.1f us (max %.1f), "
           "setup %.2f, first operands %.2f, main loop %.2f, output stage %.2f us (vector staging %.2f, compute + store issue %.2f, store retirement %.2f)\n", F8 ? "mx" : "bf16", a.M, a.N, a.K, BN, a.geglu ? (LNF ? "LN-folded GEGLU" : "GEGLU") : a.R ? "resid" : "plain", tiles, ms * 1e3 / reps,
           (t1 - t0) * tick, start / tiles * tick, startmax * tick, ph[0] / tiles * tick, ph[1] / tiles * tick, ph[2] / tiles * tick, ph[3] / tiles * tick, e5 / tiles * tick, e6 / tiles * tick, (ph[3] - e5 - e6) / tiles * tick);
}

int main(int argc, char** argv) {
    const bool f8 = argc < 2 || !strcmp(argv[1], "mx");
    const int M = argc > 2 ? atoi(argv[2]) : 4352, N = argc > 3 ? atoi(argv[3]) : 3072, K = argc > 4 ? atoi(argv[4]) : 3072, BN = argc > 5 ? atoi(argv[5]) : 224;
    const bool resid = argc > 6 && !strcmp(argv[6], "resid");
    const bool geglu = argc > 6 && !strcmp(argv[6], "geglu"), lngeglu = argc > 6 && !strcmp(argv[6], "lngeglu");
    const size_t es = f8 ? 1 : 2;
    std::vector<unsigned char> h((size_t)std::max(M, N) * K * es);
    void *A, *W, *C; uint32_t *SA, *SW; float* bias;
    CK(hipMalloc(&A, (size_t)M * K * es)); CK(hipMalloc(&W, (size_t)N * K * es)); CK(hipMalloc(&C, (size_t)M * N * 2));
    CK(hipMalloc(&SA, (size_t)(K / 128 + 1) * M * 4)); CK(hipMalloc(&SW, (size_t)(K / 128 + 1) * N * 4)); CK(hipMalloc(&bias, N * 4));
    srand(1);
    for (size_t i = 0; i < (size_t)M * K * es; ++i) h[i] = f8 ? rand() % 120 : ((i & 1) ? 0x3c + rand() % 3 : rand() & 255);       // e4m3 < 256 / bf16 around 0.01..0.1
    CK(hipMemcpy(A, h.data(), (size_t)M * K * es, hipMemcpyHostToDevice));
    for (size_t i = 0; i < (size_t)N * K * es; ++i) h[i] = f8 ? rand() % 120 : ((i & 1) ? 0x3c + rand() % 3 : rand() & 255);
    CK(hipMemcpy(W, h.data(), (size_t)N * K * es, hipMemcpyHostToDevice));
    CK(hipMemset(SA, 0x70, (size_t)(K / 128 + 1) * M * 4)); CK(hipMemset(SW, 0x70, (size_t)(K / 128 + 1) * N * 4)); CK(hipMemset(bias, 0, N * 4)); CK(hipMemset(C, 0, (size_t)M * N * 2));
    GemmArgs a; memset(&a, 0, sizeof(a));
    a.A = A; a.lda = K; a.W = W; a.M = M; a.N = N; a.K = K; a.bias = bias; a.C = C; a.ldc = N; a.rows_per_batch = M; a.splitk = 1;
    a.ep_general = getenv("LDX_EP_GENERAL") ? atoi(getenv("LDX_EP_GENERAL")) : 0;      // 1: the general output stage (what launch_gemm's switch selects)
    if (resid) { a.R = C; a.ldr = N; }
    if (geglu || lngeglu) { a.geglu = 1; a.ldc = N / 2; }
    if (lngeglu) { a.ln_c1 = bias; a.ln_eps = 1e-5f; }
    if (f8) { a.f8 = 1; a.SA = SA; a.sa_ld = M; a.SW = SW; a.sw_ld = N; }
    const int reps = 20;
    if (f8) { if (BN == 224) run<true, 224>(a, reps); else if (BN == 192) run<true, 192>(a, reps); else if (BN == 160) run<true, 160>(a, reps); else run<true, 128>(a, reps); }
    else if (lngeglu) run<false, 128, true>(a, reps);
    else { if (BN == 256) run<false, 256>(a, reps); else if (BN == 160) run<false, 160>(a, reps); else run<false, 128>(a, reps); }
    return 0;
}
