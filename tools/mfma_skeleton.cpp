// Skeleton probe, round 4: k_gemm8's phase structure (8 waves = 2 wave rows one barrier apart, 4 phases per K tile, 16 MFMAs 16x16x32 per phase per wave,
// the kernel's own operand / accumulator register pattern: 32 distinct accumulators, 16 A + 8 B fragment registers) with NO memory traffic in the loop,
// on operands loaded once from global memory: zeros, a constant, or random fp16.  Reports TFLOP/s and the effective shader clock measured in the kernel
// (s_memtime = shader cycles vs s_memrealtime = 100 MHz) - i.e. how much of the "MFMA + barriers only" gap is clock (power) and how much is the barrier
// hand-over.    hipcc --offload-arch=gfx950 -O3 tools/mfma_skeleton.cpp -o /tmp/sk && /tmp/sk
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// MODE 0: SYNC + END barrier per phase (the kernel), 1: one barrier per phase, 2: none.  BIG: 32x32x16 MFMAs (8 per phase) instead of 16x16x32 (16 per phase)
template <int MODE, bool BIG>
__global__ __launch_bounds__(512) void k(const f16* __restrict__ src, float* __restrict__ out, unsigned long long* __restrict__ clk, int ktiles) {
    const int tid = threadIdx.x, wr = tid >> 8;
    f16x8 fa[2][4][2], fb[2][2][2];                     // [A half][row tile][kk], [B half][col tile][kk]
    const f16x8* s8 = reinterpret_cast<const f16x8*>(src) + (size_t)(blockIdx.x * 512 + tid) * 24;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) fa[h][i][kk] = s8[(h * 4 + i) * 2 + kk];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) fb[h][j][kk] = s8[16 + (h * 2 + j) * 2 + kk];
    f32x4 acc[2][4][2][2];
    f32x16 accb[2][2][2];                               // BIG: [A half][row tile pair][B half] 32 x 32 tiles
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[a][i][c][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int e = 0; e < 16; ++e) accb[a][i][c][e] = 0.f;
    auto mma = [&](int ha, int hb) {
        if (BIG) {
            // 64 rows x 32 columns x K = 64 as 2 (row pairs) x 4 (k steps of 16) MFMAs 32x32x16; operands: halves of the same fragment registers
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int ip = 0; ip < 2; ++ip)
                    accb[ha][ip][hb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[hb][ks & 1][ks >> 1], fa[ha][ip * 2 + (ks & 1)][ks >> 1], accb[ha][ip][hb], 0, 0, 0);
        } else {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[ha][i][hb][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[hb][j][kk], fa[ha][i][kk], acc[ha][i][hb][j], 0, 0, 0);
        }
    };
#define PHASE(ha, hb)                                                                                                         \
    do {                                                                                                                      \
        if (MODE != 2) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_setprio(1); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } \
        mma(ha, hb);                                                                                                          \
        if (MODE == 0) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_setprio(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } \
    } while (0)
    __syncthreads();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if (MODE != 2 && wr == 1) __builtin_amdgcn_s_barrier();
    for (int t = 0; t < ktiles; ++t) { PHASE(0, 0); PHASE(0, 1); PHASE(1, 1); PHASE(1, 0); }
    if (MODE != 2 && wr == 0) __builtin_amdgcn_s_barrier();
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int j = 0; j < 2; ++j) s += acc[a][i][c][j][0] + acc[a][i][c][j][3];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int c = 0; c < 2; ++c) s += accb[a][i][c][0] + accb[a][i][c][15];
    out[blockIdx.x * 512 + tid] = s;
    if (tid == 0) { clk[blockIdx.x * 2] = c1 - c0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <typename K>
static void run(const char* label, K kern, const f16* src, float* out, unsigned long long* clk, int ktiles) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, src, out, clk, ktiles);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, src, out, clk, ktiles);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(512);
    hipMemcpy(h.data(), clk, 512 * 8, hipMemcpyDeviceToHost);
    double cyc = 0, rt = 0;
    for (int b = 0; b < 256; ++b) { cyc += h[2 * b]; rt += h[2 * b + 1]; }
    const double tf = 3.0 * 256 * 8 * (double)ktiles * 64 * (2.0 * 16 * 16 * 32) / (ms * 1e-3) / 1e12;
    const double ghz = cyc / rt * 0.1;                  // s_memrealtime ticks at 100 MHz
    printf("%-44s %7.1f TFLOP/s   clock %.3f GHz   cycles per K tile %7.1f (2048 = matrix pipe back to back)   %.0f%% of the peak at that clock\n", label, tf, ghz, cyc / 256 / ktiles,
           100.0 * tf / (2500.0 * ghz / 2.4));
}

int main() {
    const size_t n = (size_t)256 * 512 * 24 * 8;
    std::vector<f16> h(n);
    f16* src[3]; float* out; unsigned long long* clk;
    (void)hipMalloc(&out, 256L * 512 * 4); (void)hipMalloc(&clk, 512 * 8);
    for (int v = 0; v < 3; ++v) {
        srand(1234);
        for (size_t i = 0; i < n; ++i) h[i] = v == 0 ? (f16)0.f : (v == 1 ? (f16)0.5f : (f16)((rand() / (float)RAND_MAX - 0.5f) * 4.f));
        (void)hipMalloc(&src[v], n * 2); hipMemcpy(src[v], h.data(), n * 2, hipMemcpyHostToDevice);
    }
    const char* dn[3] = {"zeros", "constant 0.5", "random (-2, 2)"};
    const int KT = 20000;
    for (int v = 0; v < 3; ++v) {
        char lab[128];
        snprintf(lab, 128, "16x16x32, SYNC + END barriers, %s", dn[v]); run(lab, k<0, false>, src[v], out, clk, KT);
        snprintf(lab, 128, "16x16x32, one barrier / phase, %s", dn[v]); run(lab, k<1, false>, src[v], out, clk, KT);
        snprintf(lab, 128, "16x16x32, no barriers, %s", dn[v]); run(lab, k<2, false>, src[v], out, clk, KT);
        snprintf(lab, 128, "32x32x16, SYNC + END barriers, %s", dn[v]); run(lab, k<0, true>, src[v], out, clk, KT);
        snprintf(lab, 128, "32x32x16, one barrier / phase, %s", dn[v]); run(lab, k<1, true>, src[v], out, clk, KT);
        snprintf(lab, 128, "32x32x16, no barriers, %s", dn[v]); run(lab, k<2, true>, src[v], out, clk, KT);
    }
    return 0;
}
